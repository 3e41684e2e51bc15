#!/usr/bin/env python3
"""Randomised parity stress of the paths rebuilt this round: KMC_ALGO_SORT (MSD sort + run-length, accumulator),
AUTO with hand-overs, kmc_finalize merges, and the LR mode (rank pairs) -- many random shapes against the C oracle.
usage: python tools/stress_sort_lr.py [seconds]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
kmc = importlib.import_module("k-mer-count_amd")
import oracle_py
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "12345")))
t_end = time.time() + budget
n_cases = 0
ALG = {"sort": kmc.ALGO_SORT, "auto": kmc.ALGO_AUTO, "walk": kmc.ALGO_WALK, "stream": kmc.ALGO_STREAM}
while time.time() < t_end:
    k = int(rng.choice([5, 9, 15, 16, 21, 27, 31, 32, 33, 40, 47, 48, 55, 63]))
    pool = int(rng.choice([0, 0, 1, 3, 10, 40, 300]))
    n_rec = int(rng.integers(1, int(os.environ.get("STRESS_MAX_REC", "6000"))))
    canonical = bool(rng.integers(0, 2))
    algo = str(rng.choice(["sort", "sort", "auto", "walk", "stream"]))
    s = kmc.Synth(seed=int(rng.integers(1, 1 << 30)), pool=pool)
    hb, ho = kmc.synth_reads_host(s, int(rng.integers(0, 1000)), n_rec)
    if rng.integers(0, 3) == 0:   # ragged reads: cut every record at a random length (may be shorter than k)
        lens = rng.integers(0, 401, size=n_rec)
        pieces = [hb[int(ho[i]):int(ho[i]) + int(lens[i])] for i in range(n_rec)]
        hb = np.concatenate(pieces) if pieces else np.zeros(0, np.uint8)
        ho = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    if rng.integers(0, 4) == 0 and hb.size:   # sprinkle non-ACGT bytes
        idx = rng.integers(0, hb.size, size=max(1, hb.size // 5000))
        hb = hb.copy(); hb[idx] = ord("N")
    want = oracle_py.count_kmers(hb, ho, k, canonical, method=1)
    with kmc.KmerCounter(k=k, canonical=canonical, algo=ALG[algo]) as kc:
        cut = int(rng.integers(0, n_rec + 1))
        if rng.integers(0, 2) and 0 < cut < n_rec:   # two batches
            c0 = int(ho[cut])
            kc.add_batch(hb[:c0], ho[:cut + 1]); kc.add_batch(hb[c0:], ho[cut:] - ho[cut])
        else:
            kc.add_batch(hb, ho)
        got = kc.export()
        assert got.equals(want), ("kmers", k, pool, n_rec, canonical, algo, got.n_distinct, want.n_distinct)
        if rng.integers(0, 3) == 0:   # the ctx keeps working
            kc.reset(); kc.add_batch(hb, ho)
            assert kc.export().equals(want), ("again", k, pool, n_rec, canonical, algo)
    if rng.integers(0, 4) == 0:   # LR mode on a small ragged batch (pure ACGT)
        nr = int(rng.integers(1, 40))
        lens = rng.integers(60, 420, size=nr)
        lb = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(lens.sum()))
        if rng.integers(0, 2):   # repeats: copy a stretch around
            L = int(min(200, lb.size // 3))
            if L > 0: lb[L:2 * L] = lb[:L]
        lo = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
            kc.add_batch(lb, lo)
            assert kc.export().equals(oracle_py.count_lr(lb, lo)), ("lr", nr)
    n_cases += 1
print("stress ok:", n_cases, "cases")
