#!/usr/bin/env python3
"""Same-size batches with DIFFERENT data every time (freed device buffers get reused): would expose
stale-cache reads as well as races."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
kmc = importlib.import_module("k-mer-count_amd")
import oracle_py
fails = 0
rng = np.random.default_rng(7)
for it in range(400):
    k = int(rng.choice([17, 24, 31]))
    nreads = 300
    lens = rng.integers(0, 301, nreads) if it % 2 else np.full(nreads, 150)
    offs = np.zeros(nreads + 1, np.uint64); offs[1:] = np.cumsum(lens)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]))].copy()
    want = oracle_py.count_kmers(bases, offs, k, True)
    for algo in (kmc.ALGO_WALK, kmc.ALGO_STREAM, kmc.ALGO_WALK):
        with kmc.KmerCounter(k=k, algo=algo) as kc:
            kc.add_batch(bases, offs)
            got = kc.export()
        if not got.equals(want):
            fails += 1
            w = {(int(h), int(l)): int(c) for h, l, c in zip(want.key_hi, want.key_lo, want.count)}
            g = {(int(h), int(l)): int(c) for h, l, c in zip(got.key_hi, got.key_lo, got.count)}
            missing = [kk for kk in w if kk not in g]; extra = [kk for kk in g if kk not in w]
            diff = [(kk, w[kk], g[kk]) for kk in w if kk in g and g[kk] != w[kk]]
            print(f"MISMATCH it={it} k={k} algo={algo} distinct {want.n_distinct}/{got.n_distinct} total {want.n_total}/{got.n_total} missing {len(missing)} extra {len(extra)} diff {len(diff)}", flush=True)
print("fails", fails, flush=True)
