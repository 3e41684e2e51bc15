#!/usr/bin/env python3
"""bench.py -- whole-job k-mer counting throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N == 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json config 3, the one the metric is quoted on): the parsed form of a
10 GB synthetic FASTA with the distribution of the reference's random_fasta_generator.py:5-15
(seeded re-creation, kmc_synth_*), k = 31, canonical, resident in HBM before the timed region.
One "step" = one pass of the hot path over the resident batch: reset the count table, count
every k-mer (HIP kernel), compact + sort the table on the device; with N > 1 the per-GPU tables are
reduced over RCCL first (k-mer-count_amd/distributed.py) and every rank compacts + sorts the
partition of the global table it owns.  Weak scaling: every rank holds its own 10 GB-equivalent shard (different
records of the same seeded stream), no data-path collective while counting.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant (count) kernel: algorithmic
bytes per launch (n_bases + 8*(n_reads+1)) / its hipEvent-measured duration,
against the 8 TB/s HBM peak.  `cpu_baseline` times the CPU oracle (a port of the reference's
algorithm; the Rust reference cannot be built here) on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.1-6.3 TB/s is achievable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps (the first ~3 run ~8%% slower: clock ramp-up)")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--fasta-bytes", type=float, default=10e9, help="size of the synthetic FASTA text per GPU (weak scaling)")
    ap.add_argument("--total-fasta-bytes", type=float, default=0.0,
                    help="strong scaling: ONE synthetic FASTA of this size, its records split over the ranks "
                         "(BASELINE.json config 4: --gpus 8 --total-fasta-bytes 50e9 --seed 3); overrides --fasta-bytes")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--pool", type=int, default=10)
    ap.add_argument("--algo", default="auto", choices=["auto", "stream", "walk", "sort"])
    ap.add_argument("--forward", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the untimed cold-memo steps after the timed region (profiling runs)")
    ap.add_argument("--reduce-finalize-every", type=int, default=1,
                    help="N > 1: compact + sort the owned partition (one host synchronisation) only every n-th step; "
                         "the other steps queue count -> pack -> all-gather -> merge and return (default 1: every step delivers a sorted table)")
    ap.add_argument("--no-exact-check", action="store_true", help="skip the exact full-size table check after the timed region")
    ap.add_argument("--cpu-sample-records", type=int, default=5_000_000, help="bounded CPU-baseline sample (~10-15 s of one core)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    kmc = importlib.import_module("k-mer-count_amd")
    kdist = importlib.import_module("k-mer-count_amd.distributed")
    kmc.lib()  # fail loudly if the HIP extension is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # KMC_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a
    # one-GPU box; RCCL refuses two ranks on one device).  The driver's multi-GPU run uses nccl.
    backend = os.environ.get("KMC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- resident input: this rank's shard of the seeded record stream -------------------------
    synth = kmc.Synth(seed=args.seed, pool=args.pool)
    strong = args.total_fasta_bytes > 0
    if strong:
        n_all, fasta_bytes = kmc.synth_records_for_bytes(synth, int(args.total_fasta_bytes))
        first, n_rec = kdist.shard_range(n_all, rank, world)   # contiguous, balanced record shards
    else:
        n_rec, fasta_bytes = kmc.synth_records_for_bytes(synth, int(args.fasta_bytes))
        first = rank * n_rec
        n_all = n_rec * world
    read_len = synth.read_len
    n_bases = n_rec * read_len
    d_bases = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
    d_offs = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    kmc.synth_reads_device(synth, first, n_rec, d_bases.data_ptr(), d_offs.data_ptr(), device=local_rank)
    torch.cuda.synchronize()
    k = args.k
    n_kmers = n_rec * (read_len - k + 1)
    n_kmers_all = n_all * (read_len - k + 1)   # all ranks together
    algo = {"auto": kmc.ALGO_AUTO, "stream": kmc.ALGO_STREAM, "walk": kmc.ALGO_WALK, "sort": kmc.ALGO_SORT}[args.algo]

    # N > 1: both ctxs queue their kernels on ONE torch stream, the stream the RCCL collective is
    # ordered against, so count -> pack -> all-gather -> merge needs no host synchronisation
    # (the library's hipEvents still bracket the count kernel on that stream).
    side = torch.cuda.Stream(dev) if world > 1 else None
    sh = side.cuda_stream if side is not None else None
    kc = kmc.KmerCounter(k=k, canonical=not args.forward, device=local_rank, algo=algo, stream=sh)
    owner = kmc.KmerCounter(k=k, canonical=not args.forward, device=local_rank, stream=sh) if world > 1 else None

    kernel_ms, launches = [], []

    step_no = [0]

    def step(record):
        step_no[0] += 1
        kc.reset()
        kc.add_batch_device(d_bases.data_ptr(), d_offs.data_ptr(), n_rec, n_bases, read_len)
        if world > 1:
            owner.reset()
            with torch.cuda.stream(side):
                # pack the live table, one all-gather of slabs, owner merge, owner finalize (compact + sort):
                # the product of a step is the owner-partitioned sorted table; checked after the timed region
                fin = step_no[0] % max(args.reduce_finalize_every, 1) == 0
                _, nd = kdist.reduce_tables(kc, owner, report_sent=False, finalize=fin)
        else:
            nd, nt = kc.finalize()
            if nt != n_kmers:
                raise SystemExit(f"count mismatch: table sums to {nt}, expected {n_kmers}")
        if record:
            st = kc.stats()
            kernel_ms.append(st.kernel_ms_last)
            launches.append(st.launches_last)
        return nd

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nd = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = kc.stats()
    algo_used = {1: "stream", 2: "walk", 3: "sort"}.get(st.algo_last, "?")
    # Outside the timed region, for transparency: the same step with the walk kernel's memo dropped
    # first (kmc_forget_source).  The memo is graph STRUCTURE learned from earlier launches (no
    # counts); timed steps reuse it, as every batch after the first of a real file does.
    cold_ms = []
    if algo_used == "walk" and not args.no_cold:
        for _ in range(3):
            kc.forget_source(memo=True, history=False)
            step(False)
            cold_ms.append(kc.stats().kernel_ms_last)
    # ---- exact result check, outside the timed region (checker: tests/analytic_oracle.py) ----------
    # The table this rank counted in the last step must equal, key for key and count for count, the
    # exact table of its whole record range (line / adjacent-pair histograms over ALL records expanded
    # through the pool; a few seconds of host time at 22 M records) -- not a prefix, not a bound.
    exact_full = None
    if args.pool > 0 and not args.no_exact_check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import analytic_oracle  # test infrastructure, used here only as the checker
        tx = time.perf_counter()
        want_full = analytic_oracle.exact_table(args.seed, k, first, n_rec, canonical=not args.forward, pool=args.pool)
        got_full = kc.export()
        exact_full = {"bit_exact": bool(got_full.equals(want_full)), "records": n_rec, "distinct": want_full.n_distinct,
                      "kmers": want_full.n_total, "host_s": round(time.perf_counter() - tx, 2)}
        if not exact_full["bit_exact"]:
            raise SystemExit(f"rank {rank}: GPU table differs from the exact table of records [{first}, {first + n_rec})")
    reduced = None
    if world > 1:
        # outside the timed region: the owner-partitioned result of the last step must account for
        # every k-mer of every rank, and owners must not overlap (distinct keys add up)
        own_nd, own_nt = owner.finalize()
        t = torch.tensor([own_nd, own_nt], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        reduced = {"distinct_all_owners": int(t[0].item()), "kmers_all_owners": int(t[1].item())}
        if reduced["kmers_all_owners"] != n_kmers_all:
            raise SystemExit(f"reduce mismatch: owners hold {reduced['kmers_all_owners']} k-mers, expected {n_kmers_all}")

    # ---- roofline of the dominant kernel (this rank's launches; every rank runs the same shape) --
    algo_bytes = n_bases + 8 * (n_rec + 1)  # 1 B/base ASCII + the offsets array
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes of tools/profile_bench.sh (FETCH_SIZE / WRITE_SIZE in their
    # own rocprofv3 runs, gfx950 correction applied by tools/summarize_profile.py); only reported when a
    # committed profile matches this exact workload and algorithm
    traffic = None
    import glob
    for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            pj = json.load(open(prof))
            if pj.get("workload_bases") == n_bases and pj.get("k") == k and pj.get("algo") == algo_used:
                traffic = pj.get("hbm_bytes_per_launch")
                break
        except Exception:
            pass
    dominant = {"walk": "kmc_walk_kernel", "stream": "kmc_stream_kernel",
                "sort": "sort pipeline: kmc_stream_kernel<SINK=1> (extract) + kmc_msd_{hist,scan,scatter,leaf,gather}_kernel"}.get(algo_used, "?")
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": dominant, "kernel_ms": round(k_ms, 4),
                "launches_per_step": int(round(float(np.mean(launches)))) if launches else None,
                "kernel_ms_cold_memo": round(float(np.mean(cold_ms)), 4) if cold_ms else None,
                "algorithmic_bytes_per_step": algo_bytes}
    if algo_used == "sort":
        # The sort path's own traffic: one key per base position out of the extraction, two
        # levels of histogram read + scatter read/write, the leaves (read, staged keys + counts written), the
        # gather (both read, both written): n_bases + 13 n_kmers key-units of 8 B (k <= 31) or 16 B.  `frac`
        # above prices the INPUT bytes (the metric's definition); this prices the pipeline against HBM.
        kb = 8 if k <= 31 else 16
        model = (n_bases + 13 * n_kmers) * kb
        roofline["sort_pipeline"] = {"model_bytes_per_step": model, "achieved": round(model / (k_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                                     "frac": round(model / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "kernels": "kmc_stream_kernel<SINK=1>, kmc_msd_{hist,scan_a,scan,scatter,leaf,gather}_kernel"}

    # ---- CPU baseline: the oracle (port of the reference's algorithm), bounded sample ----------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_py  # the checker/baseline, never the product
        ns = min(args.cpu_sample_records, n_rec)
        hb, ho = kmc.synth_reads_host(synth, first, ns)
        tc = time.perf_counter()
        want = oracle_py.count_kmers(hb, ho, k, canonical=not args.forward, method=1)
        cpu_s = time.perf_counter() - tc
        with kmc.KmerCounter(k=k, canonical=not args.forward, device=local_rank, algo=algo) as kv:
            kv.add_batch(hb, ho)
            exact = bool(kv.export().equals(want))
        cpu = {"value": round(want.n_total / cpu_s, 1), "unit": "k-mers/s", "cores": 1, "kind": "port",
               "sample": f"first {ns} records ({ns * read_len} bases) of the same workload; oracle hash-map counter "
                         f"(2-bit rolling encode + canonical + open addressing), {cpu_s:.2f} s",
               "gpu_bit_exact_on_sample": exact}
        if not exact:
            raise SystemExit("GPU table differs from the CPU oracle on the baseline sample")
        # SURVEY.md 8d's other two CPU views (reported next to the baseline, not instead of it):
        # B1 = the reference's algorithm SHAPE on contiguous k (materialise every window as a string,
        # comparison sort, run-length: main.rs:78-79,87), 1 core like the reference;
        # B2 on all host cores = the same hash-map counter, one record shard per thread (tables not merged).
        extra = {}
        nb1 = min(20_000, n_rec)
        tb = time.perf_counter()
        w1 = oracle_py.count_kmers_strings(hb[:nb1 * read_len], ho[:nb1 + 1], k, canonical=not args.forward)
        dt1 = time.perf_counter() - tb
        extra["b1_reference_shape"] = {"value": round(w1.n_total / dt1, 1), "unit": "k-mers/s", "cores": 1, "kind": "port",
                                       "sample": f"first {nb1} records; every window materialised as a string, qsort, run-length; {dt1:.2f} s"}
        import threading
        T = max(1, min(os.cpu_count() or 1, 16))
        per = max(1, min(600_000, ns // T))
        totals = [0] * T

        def shard(i):
            b0 = i * per * read_len
            o = (ho[i * per:(i + 1) * per + 1] - ho[i * per]).copy()
            totals[i] = oracle_py.count_kmers(hb[b0:b0 + per * read_len], o, k, canonical=not args.forward, method=1).n_total

        th = [threading.Thread(target=shard, args=(i,)) for i in range(T)]
        tb = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        dt2 = time.perf_counter() - tb
        extra["b2_all_cores"] = {"value": round(sum(totals) / dt2, 1), "unit": "k-mers/s", "cores": T, "kind": "port",
                                 "sample": f"{T} threads x {per} records of the same workload, one hash-map table per thread (not merged); {dt2:.2f} s"}
        cpu["other_views"] = extra

    if rank == 0:
        total_kmers = n_kmers_all * args.steps
        out = {
            "metric": "k-mers/sec (whole node) on synthetic FASTA, k=%d; counts bit-exact vs ref" % k,
            "value": round(total_kmers / elapsed, 1),
            "unit": "k-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%.0f GB synthetic FASTA%s (random_fasta_generator.py distribution, seed %d, pool %d), "
                                   "k=%d, %s, %dxMI355X" % (fasta_bytes / 1e9, " in total, records split over the ranks" if strong else (" per GPU" if world > 1 else ""),
                                                            args.seed, args.pool, k, "forward" if args.forward else "canonical", world),
                       "records_per_gpu": n_rec, "bases_per_gpu": n_bases, "kmers_per_gpu": n_kmers,
                       "distinct": int(reduced["distinct_all_owners"]) if reduced else int(nd), "algo": algo_used, "sharding": "records, one shard per GPU; RCCL table reduce (one all-gather of fixed-size slabs)"
                       if world > 1 else "single GPU", "reduced": reduced, "exact_full_size_check": exact_full},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
