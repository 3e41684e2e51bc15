#!/usr/bin/env python3
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
kmc = importlib.import_module("k-mer-count_amd")
import oracle_py
fails = 0
for k in (24, 17, 30, 31, 7):
    rng = np.random.default_rng(100 + k)
    lens = rng.integers(0, 301, 300)
    offs = np.zeros(301, np.uint64); offs[1:] = np.cumsum(lens)
    n = int(offs[-1])
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
    for canonical in (True, False):
        want = oracle_py.count_kmers(bases, offs, k, canonical)
        for rep in range(60):
            for algo in (kmc.ALGO_STREAM, kmc.ALGO_AUTO, kmc.ALGO_WALK):
                with kmc.KmerCounter(k=k, canonical=canonical, algo=algo) as kc:
                    kc.add_batch(bases, offs)
                    got = kc.export()
                    st = kc.stats()
                if not got.equals(want):
                    fails += 1
                    w = {(int(h), int(l)): int(c) for h, l, c in zip(want.key_hi, want.key_lo, want.count)}
                    g = {(int(h), int(l)): int(c) for h, l, c in zip(got.key_hi, got.key_lo, got.count)}
                    missing = [kk for kk in w if kk not in g]; extra = [kk for kk in g if kk not in w]
                    diff = [(kk, w[kk], g[kk]) for kk in w if kk in g and g[kk] != w[kk]]
                    print(f"MISMATCH k={k} canon={canonical} rep={rep} algo={algo} used={st.algo_last} distinct {want.n_distinct}/{got.n_distinct} total {want.n_total}/{got.n_total} missing {len(missing)} extra {len(extra)} diff {len(diff)} kmers_stat {st.n_kmers}", flush=True)
                    for kk in missing[:4]: print("  missing", kk, w[kk])
                    for kk in extra[:4]: print("  extra", kk, g[kk])
                    for d in diff[:4]: print("  diff", d)
print("fails", fails, flush=True)
