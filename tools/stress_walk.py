#!/usr/bin/env python3
"""Repeats KMC_ALGO_WALK on inputs that overflow its memo tables and diffs against the oracle."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
kmc = importlib.import_module("k-mer-count_amd")
import oracle_py
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 124)
fails = 0
for trial in range(40):
    k = int(rng.choice([5, 16, 24, 31]))
    nreads = int(rng.choice([50, 300, 3000]))
    lens = rng.integers(0, int(rng.choice([60, 300, 416])) + 1, nreads)
    offs = np.zeros(nreads + 1, np.uint64); offs[1:] = np.cumsum(lens)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]))].copy()
    want = oracle_py.count_kmers(bases, offs, k, True)
    for rep in range(4):
        with kmc.KmerCounter(k=k, algo=kmc.ALGO_WALK) as kc:
            kc.add_batch(bases, offs)
            got = kc.export()
        if not got.equals(want):
            fails += 1
            w = {(int(h), int(l)): int(c) for h, l, c in zip(want.key_hi, want.key_lo, want.count)}
            g = {(int(h), int(l)): int(c) for h, l, c in zip(got.key_hi, got.key_lo, got.count)}
            missing = [(kk, w[kk]) for kk in w if kk not in g]
            extra = [(kk, g[kk]) for kk in g if kk not in w]
            diff = [(kk, w[kk], g[kk]) for kk in w if kk in g and g[kk] != w[kk]]
            print(f"MISMATCH trial {trial} rep {rep} k={k} reads={nreads} bases={int(offs[-1])} distinct {want.n_distinct} got {got.n_distinct} total {want.n_total} got {got.n_total}: "
                  f"missing {len(missing)} extra {len(extra)} diffcount {len(diff)}", flush=True)
            for kk, c in (missing[:3]):
                print("   missing", kmc.Table(np.array([kk[0]], np.uint64), np.array([kk[1]], np.uint64), np.array([c], np.uint64), k).to_bytes().decode().strip())
            for kk, c in (extra[:3]):
                print("   extra  ", kmc.Table(np.array([kk[0]], np.uint64), np.array([kk[1]], np.uint64), np.array([c], np.uint64), k).to_bytes().decode().strip())
            for kk, a, b in diff[:3]:
                print("   count  ", kk, a, b)
print("fails", fails, flush=True)
