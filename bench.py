#!/usr/bin/env python3
"""bench.py -- whole-job k-mer counting throughput on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N == 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json; input = the parsed form of a synthetic FASTA with the distribution of the reference's
random_fasta_generator.py:5-15, seeded re-creation kmc_synth_*, resident in HBM before the timed region):
  N == 1   config 3, the one the metric is quoted on: 10 GB, seed 2, k = 31, canonical.
  N  > 1   config 4 AS STATED: ONE 50 GB file, seed 3, k = 31, its records split over the N GPUs (records are the
           reference's unit of work, k-mer-count/src/main.rs:58-62,73-75), per-GPU count tables reduced over RCCL:
           "scaling": "strong".  The weak-scaling figure (10 GB per GPU, seed 2 -- config 3 on every rank) is measured
           right after it and reported in the same line as `weak_scaling`.
  --fasta-bytes B (per GPU, weak) or --total-fasta-bytes B (strong) select one workload explicitly.
One "step" = one pass of the hot path over the resident batch: reset the count table, count every k-mer (HIP
kernel), compact + sort the table on the device; with N > 1 the per-GPU tables are reduced over RCCL first
(k-mer-count_amd/distributed.py) and every rank compacts + sorts the partition of the global table it owns.  No
data-path collective while counting.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant (count) kernel: algorithmic bytes per launch
(n_bases + 8*(n_reads+1)) / its hipEvent-measured duration, against the nominal 8 TB/s HBM peak AND
against the streaming-read rate measured in this run by a plain read-only kernel over the same bytes
(kmc_read_peak_device).  `cpu_baseline` times the CPU oracle (a port of the reference's algorithm; the Rust reference
cannot be built here) on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import glob
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E nominal peak (MI355X_MICROARCH.md); what a plain read reaches is measured below


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps (the first ~3 run ~8%% slower: clock ramp-up)")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--fasta-bytes", type=float, default=None, help="size of the synthetic FASTA text per GPU (weak scaling; default at N=1: 10e9)")
    ap.add_argument("--total-fasta-bytes", type=float, default=None,
                    help="strong scaling: ONE synthetic FASTA of this size, its records split over the ranks "
                         "(default at N>1: BASELINE.json config 4 = 50e9, seed 3); overrides --fasta-bytes")
    ap.add_argument("--seed", type=int, default=None, help="default: 2 (config 3), 3 for the N>1 default (config 4)")
    ap.add_argument("--pool", type=int, default=10)
    ap.add_argument("--algo", default="auto", choices=["auto", "stream", "walk", "sort"])
    ap.add_argument("--forward", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the untimed cold-memo steps after the timed region (profiling runs)")
    ap.add_argument("--no-weak", action="store_true", help="N>1 default run: skip the weak-scaling sub-measurement")
    ap.add_argument("--no-read-peak", action="store_true", help="skip the measured streaming-read peak (kmc_read_peak_device)")
    ap.add_argument("--reduce-finalize-every", type=int, default=5,
                    help="N > 1: the host looks at the owner's result (one synchronisation + the every-step-delivered check) only "
                         "every n-th step; EVERY step queues count -> pack -> all-gather -> merge -> finalize of the owned partition "
                         "(kmc_finalize_async: the sorted table is produced on the device each step).  1: synchronise every step")
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

    k = args.k
    canonical = not args.forward
    algo = {"auto": kmc.ALGO_AUTO, "stream": kmc.ALGO_STREAM, "walk": kmc.ALGO_WALK, "sort": kmc.ALGO_SORT}[args.algo]
    # N > 1: both ctxs queue their kernels on ONE torch stream, the stream the RCCL collective is
    # ordered against, so count -> pack -> all-gather -> merge -> finalize needs no host synchronisation
    # (the library's hipEvents still bracket the count kernel on that stream).
    side = torch.cuda.Stream(dev) if world > 1 else None
    sh = side.cuda_stream if side is not None else None
    kc = kmc.KmerCounter(k=k, canonical=canonical, device=local_rank, algo=algo, stream=sh)
    owner = kmc.KmerCounter(k=k, canonical=canonical, device=local_rank, stream=sh) if world > 1 else None
    every = max(args.reduce_finalize_every, 1)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(seed, strong, size_bytes, first_run):
        """One workload: resident input, W untimed + K timed steps, checks.  Returns the numbers of the timed region."""
        synth = kmc.Synth(seed=seed, pool=args.pool)
        if strong:
            n_all, fasta_bytes = kmc.synth_records_for_bytes(synth, int(size_bytes))
            first, n_rec = kdist.shard_range(n_all, rank, world)   # contiguous, balanced record shards
        else:
            n_rec, fasta_bytes = kmc.synth_records_for_bytes(synth, int(size_bytes))
            first = rank * n_rec
            n_all = n_rec * world
        read_len = synth.read_len
        n_bases = n_rec * read_len
        d_bases = torch.empty(n_bases + 64, dtype=torch.uint8, device=dev)
        d_offs = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
        kmc.synth_reads_device(synth, first, n_rec, d_bases.data_ptr(), d_offs.data_ptr(), device=local_rank)
        torch.cuda.synchronize()
        n_kmers = n_rec * (read_len - k + 1)
        n_kmers_all = n_all * (read_len - k + 1)   # all ranks together
        if not first_run:   # a second workload on the same ctxs: a different pool -- start from what a fresh ctx knows
            kc.forget_source(memo=True, history=True)
        pb, po = d_bases.data_ptr(), d_offs.data_ptr()
        seen = {"ok": None, "skip": None, "steps": 0}

        def check_owner():
            """(after a synchronising call on `owner`) every step since the last look has delivered its sorted partition
            through the small-table path and no slab was skipped -- else this is not the measurement it claims to be"""
            st = owner.stats()
            if seen["ok"] is not None:
                if st.n_async_ok - seen["ok"] != seen["steps"] or st.n_async_slabs_skipped != seen["skip"]:
                    raise SystemExit(f"rank {rank}: {seen['steps']} steps queued, {st.n_async_ok - seen['ok']} delivered a sorted partition, "
                                     f"{st.n_async_slabs_skipped - seen['skip']} oversize slabs: rerun with --reduce-finalize-every 1")
            seen.update(ok=st.n_async_ok, skip=st.n_async_slabs_skipped, steps=0)

        def step(i, last):
            kc.reset()
            kc.add_batch_device(pb, po, n_rec, n_bases, read_len)
            if world > 1:
                owner.reset()
                with torch.cuda.stream(side):
                    # pack the live table, one all-gather of slabs, owner merge, owner finalize (compact + sort): the product of
                    # a step is the owner-partitioned sorted table, produced on the device EVERY step; the host looks at it
                    # (sizes, delivery check) every `every`-th step and at the last one
                    fin = last or (i + 1) % every == 0
                    _, nd = kdist.reduce_tables(kc, owner, report_sent=False, finalize=fin)
                    seen["steps"] += 1
                    if fin:
                        check_owner()
                return nd
            nd, nt = kc.finalize()
            if nt != n_kmers:
                raise SystemExit(f"count mismatch: table sums to {nt}, expected {n_kmers}")
            return nd

        # the plain read kernel runs once in front of everything (it also brings a fresh box's clocks up: on some boxes of the
        # pool the first ~40 ms of work run 3-5 % slower, more than five warm-up steps cover) and again behind the timed region
        nb16 = (n_bases // 16) * 16

        def read_peak(best):
            for shape in (0, 1, 2, 3):
                ms, _ = kmc.read_peak_device(pb, nb16, device=local_rank, stream=0, shape=shape, iters=5)
                gbs = nb16 / (ms * 1e-3) / 1e9
                if best is None or gbs > best[0]:
                    best = (gbs, shape, ms)
            return best

        best_before = read_peak(None) if (not args.no_read_peak and first_run) else None
        if world > 1:
            check_owner()   # (baseline: nothing is queued on `owner` at this point, its stats are current)
        for i in range(args.warmup):
            step(i, i == args.warmup - 1)
        fence()
        s0 = kc.stats()
        t0 = time.perf_counter()
        nd = None
        for i in range(args.steps):
            r = step(i, i == args.steps - 1)
            nd = r if r is not None else nd
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        s1 = kc.stats()   # (reads the launch events of every batch since s0: kernel_ms_lifetime survives kmc_reset)
        n_l = max(int(s1.launches_lifetime - s0.launches_lifetime), 1)
        k_ms = (s1.kernel_ms_lifetime - s0.kernel_ms_lifetime) / args.steps
        algo_used = {1: "stream", 2: "walk", 3: "sort"}.get(s1.algo_last, "?")
        # Outside the timed region, for transparency: the same step with the walk kernel's memo dropped
        # first (kmc_forget_source).  The memo is graph STRUCTURE learned from earlier launches (no
        # counts); timed steps reuse it, as every batch after the first of a real file does.
        cold_ms = []
        if algo_used == "walk" and not args.no_cold and first_run:
            for _ in range(3):
                kc.forget_source(memo=True, history=False)
                step(0, True)
                torch.cuda.synchronize()
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
            want_full = analytic_oracle.exact_table(seed, k, first, n_rec, canonical=canonical, pool=args.pool)
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
        # ---- the streaming-read rate this GPU reaches on these very bytes (plain read-only kernel, no product code) ----
        peak = None
        if not args.no_read_peak and first_run:
            best = read_peak(best_before)
            peak = {"GBps": round(best[0], 1), "ms": round(best[2], 4), "bytes": nb16, "kernel": "kmc_read_peak_kernel, " + PEAK_SHAPES[best[1]],
                    "how": "nt dwordx4 loads + xor-reduce over the resident batch, best of 4 grid shapes, 5 launches each after a warm-up, one hipEvent pair; "
                           "measured in front of the warm-up steps and again behind the timed region, the faster of the two"}
        res = {"seed": seed, "strong": strong, "fasta_bytes": fasta_bytes, "n_rec": n_rec, "n_all": n_all, "n_bases": n_bases, "n_kmers": n_kmers,
               "n_kmers_all": n_kmers_all, "elapsed": elapsed, "kernel_ms": k_ms, "launches_per_step": n_l / args.steps, "algo_used": algo_used,
               "cold_ms": cold_ms, "exact_full": exact_full, "reduced": reduced, "nd": nd, "read_peak": peak, "read_len": read_len,
               "first": first, "synth": synth}
        del d_bases, d_offs
        torch.cuda.empty_cache()
        return res

    explicit = args.fasta_bytes is not None or args.total_fasta_bytes is not None
    if args.total_fasta_bytes is not None:
        main_cfg = dict(seed=args.seed if args.seed is not None else 2, strong=True, size_bytes=args.total_fasta_bytes)
    elif args.fasta_bytes is not None or world == 1:
        main_cfg = dict(seed=args.seed if args.seed is not None else 2, strong=False, size_bytes=args.fasta_bytes if args.fasta_bytes is not None else 10e9)
    else:   # N > 1, nothing chosen: BASELINE.json config 4 as stated
        main_cfg = dict(seed=args.seed if args.seed is not None else 3, strong=True, size_bytes=50e9)
    m = measure(first_run=True, **main_cfg)
    weak = None
    if world > 1 and not explicit and not args.no_weak:
        weak = measure(first_run=False, seed=2, strong=False, size_bytes=10e9)

    # ---- roofline of the dominant kernel (this rank's launches; every rank runs the same shape) --
    n_bases, n_rec, algo_used = m["n_bases"], m["n_rec"], m["algo_used"]
    algo_bytes = n_bases + 8 * (n_rec + 1)  # 1 B/base ASCII + the offsets array
    k_ms = m["kernel_ms"]
    achieved = algo_bytes / (k_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes of tools/profile_bench.sh (FETCH_SIZE / WRITE_SIZE in their
    # own rocprofv3 runs, gfx950 correction applied by tools/summarize_profile.py); only reported when a
    # committed profile matches this exact workload and algorithm -- NOT measured in this run (traffic_source says so)
    traffic, traffic_source = None, None
    for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            pj = json.load(open(prof))
            if pj.get("workload_bases") == n_bases and pj.get("k") == k and pj.get("algo") == algo_used:
                traffic = pj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/%s (rocprofv3 --pmc passes of the same workload and kernel on an earlier run; not measured in this run)" % os.path.basename(prof)
                break
        except Exception:
            pass
    dominant = {"walk": "kmc_walk_kernel", "stream": "kmc_stream_kernel",
                "sort": "sort pipeline: kmc_stream_kernel<SINK=1> (extract) + kmc_msd_*_kernel"}.get(algo_used, "?")
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": dominant, "kernel_ms": round(k_ms, 4),
                "launches_per_step": round(m["launches_per_step"], 2),
                "kernel_ms_cold_memo": round(float(np.mean(m["cold_ms"])), 4) if m["cold_ms"] else None,
                "algorithmic_bytes_per_step": algo_bytes}
    if m["read_peak"]:
        roofline["peak_measured"] = m["read_peak"]["GBps"]
        roofline["frac_of_measured"] = round(achieved / m["read_peak"]["GBps"], 4)
        roofline["peak_measured_by"] = m["read_peak"]
    if algo_used == "sort":
        # The sort path's own traffic.  `frac` above prices the INPUT bytes (the metric's definition);
        # this prices the pipeline's modelled HBM traffic against the peak.
        kb = 8 if k <= 31 else 16
        model = int((n_bases + SORT_MODEL_KEY_UNITS * m["n_kmers"]) * kb)
        roofline["sort_pipeline"] = {"model_bytes_per_step": model, "key_units_per_kmer": SORT_MODEL_KEY_UNITS,
                                     "achieved": round(model / (k_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                                     "frac": round(model / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "model": SORT_MODEL_TEXT}

    # ---- CPU baseline: the oracle (port of the reference's algorithm), bounded sample ----------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_py  # the checker/baseline, never the product
        synth, first, read_len = m["synth"], m["first"], m["read_len"]
        ns = min(args.cpu_sample_records, n_rec)
        hb, ho = kmc.synth_reads_host(synth, first, ns)
        tc = time.perf_counter()
        want = oracle_py.count_kmers(hb, ho, k, canonical=canonical, method=1)
        cpu_s = time.perf_counter() - tc
        with kmc.KmerCounter(k=k, canonical=canonical, device=local_rank, algo=algo) as kv:
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
        w1 = oracle_py.count_kmers_strings(hb[:nb1 * read_len], ho[:nb1 + 1], k, canonical=canonical)
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
            totals[i] = oracle_py.count_kmers(hb[b0:b0 + per * read_len], o, k, canonical=canonical, method=1).n_total

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

    def workload_text(r):
        return ("%.0f GB synthetic FASTA%s (random_fasta_generator.py distribution, seed %d, pool %d), k=%d, %s, %dxMI355X"
                % (r["fasta_bytes"] / 1e9, " in total, records split over the ranks" if r["strong"] else (" per GPU" if world > 1 else ""),
                   r["seed"], args.pool, k, "canonical" if canonical else "forward", world))

    if rank == 0:
        total_kmers = m["n_kmers_all"] * args.steps
        reduced = m["reduced"]
        cfg = {"workload": workload_text(m),
               "records_per_gpu": n_rec, "bases_per_gpu": n_bases, "kmers_per_gpu": m["n_kmers"], "kmers_all_gpus": m["n_kmers_all"],
               "distinct": int(reduced["distinct_all_owners"]) if reduced else int(m["nd"]), "algo": algo_used,
               "sharding": "records, one shard per GPU; RCCL table reduce (one all-gather of fixed-size slabs)" if world > 1 else "single GPU",
               "reduced": reduced, "exact_full_size_check": m["exact_full"]}
        if world > 1:
            cfg["reduce_finalize_every"] = every
            cfg["every_step_delivers"] = ("every step queues count -> pack -> all-gather -> merge -> finalize of the owned partition (sorted table on the "
                                          "device each step); the host synchronises and checks n_async_ok / n_async_slabs_skipped every %d steps and at the last step" % every)
        out = {
            "metric": "k-mers/sec (whole node) on synthetic FASTA, k=%d; counts bit-exact vs ref" % k,
            "value": round(total_kmers / m["elapsed"], 1),
            "unit": "k-mers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(m["elapsed"] / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if m["strong"] else "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": cfg,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if weak is not None:
            wk = weak["n_kmers_all"] * args.steps
            wb = weak["n_bases"] + 8 * (weak["n_rec"] + 1)
            out["weak_scaling"] = {"workload": workload_text(weak), "scaling": "weak", "value": round(wk / weak["elapsed"], 1), "unit": "k-mers/s",
                                   "ms_per_step": round(weak["elapsed"] / args.steps * 1e3, 4), "steps": args.steps, "warmup": args.warmup,
                                   "kernel_ms": round(weak["kernel_ms"], 4), "roofline_frac": round(wb / (weak["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "kmers_per_gpu": weak["n_kmers"], "reduced": weak["reduced"], "exact_full_size_check": weak["exact_full"],
                                   "measured": "right after the strong-scaling steps above, same processes and ctxs (planner history and walk memo dropped in between)"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


# The sort pipeline's modelled HBM traffic per k-mer, in key units (8 B for k <= 31, 16 B above), DESIGN.md 4.3.
PEAK_SHAPES = {0: "256 x 1024 threads (the walk kernel's grid), 5 x 16 B loads in flight per lane",
               1: "2048 x 256 threads, 8 x 16 B", 2: "512 x 1024 threads, 4 x 16 B", 3: "1024 x 512 threads, 8 x 16 B"}
SORT_MODEL_KEY_UNITS = 8
SORT_MODEL_TEXT = ("extraction writes one key per base position (its level-0 histogram is built on the way); level 0: scatter read + write; "
                   "level 1: histogram read + scatter read + write; leaves: read, (key, count) pairs written in place")


if __name__ == "__main__":
    main()
