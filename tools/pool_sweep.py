#!/usr/bin/env python3
"""Cardinality sweep: the benchmark workload with the generator's pool size varied from the
reference's 10 lines (random_fasta_generator.py:5) to "every line fresh random" (pool 0), i.e. from
3 k distinct k-mers to almost one per window.  One JSON line per (k, pool): whole-step time, count
kernel time, algorithm AUTO ended up on, share of k-mers counted directly with global atomics.

    python tools/pool_sweep.py [--fasta-bytes 1e9] [--pools 10,16,20,26,32,50,100,1000,0] [--ks 31,63] [--algo auto]
"""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fasta-bytes", type=float, default=1e9)
    ap.add_argument("--pools", default="10,16,20,26,32,50,100,1000,0")
    ap.add_argument("--ks", default="31,63")
    ap.add_argument("--algo", default="auto")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--seed", type=int, default=2)
    args = ap.parse_args()
    import torch
    kmc = importlib.import_module("k-mer-count_amd")
    algo = {"auto": kmc.ALGO_AUTO, "stream": kmc.ALGO_STREAM, "walk": kmc.ALGO_WALK, "sort": kmc.ALGO_SORT}[args.algo]
    names = {1: "stream", 2: "walk", 3: "sort"}
    for k in [int(x) for x in args.ks.split(",")]:
        for pool in [int(x) for x in args.pools.split(",")]:
            s = kmc.Synth(seed=args.seed, pool=pool)
            n, _ = kmc.synth_records_for_bytes(s, int(args.fasta_bytes))
            d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda")
            d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
            kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr())
            torch.cuda.synchronize()
            n_kmers = n * (400 - k + 1)
            with kmc.KmerCounter(k=k, algo=algo) as kc:
                rows = []
                for step in range(args.steps):
                    kc.reset()
                    t0 = time.perf_counter()
                    kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400)
                    nd, nt = kc.finalize()
                    dt = time.perf_counter() - t0
                    st = kc.stats()
                    assert nt == n_kmers, (nt, n_kmers)
                    rows.append((dt, st.kernel_ms_last, st.algo_last, st.launches_last, st.n_direct, nd))
                dt, kms, al, nl, ndir, nd = min(rows[1:], key=lambda r: r[0])  # (step 0 learns the input: memo, history)
                first = rows[0]
                print(json.dumps({"k": k, "pool": pool, "algo_requested": args.algo, "algo_last": names.get(al, "?"),
                                  "records": n, "kmers": n_kmers, "distinct": nd,
                                  "step_ms": round(dt * 1e3, 3), "count_kernels_ms": round(kms, 3), "launches": nl,
                                  "direct_share": round(ndir / n_kmers, 4),
                                  "gkmers_per_s": round(n_kmers / dt / 1e9, 2),
                                  "first_step_ms": round(first[0] * 1e3, 3), "first_step_algo": names.get(first[2], "?")}), flush=True)
            del d_b, d_o
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
