#!/usr/bin/env python3
"""Same-size batches with DIFFERENT data every time (freed device buffers get reused): would expose
stale-cache reads as well as races.  Prints what differs."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
kmc = importlib.import_module("k-mer-count_amd")
import oracle_py
fails = 0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
names = {1: "stream", 2: "walk", 3: "sort", 0: "auto"}
for it in range(iters):
    k = int(rng.choice([17, 24, 31, 47]))
    nreads = 300
    lens = rng.integers(0, 301, nreads) if it % 2 else np.full(nreads, 150)
    offs = np.zeros(nreads + 1, np.uint64); offs[1:] = np.cumsum(lens)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]))].copy()
    want = oracle_py.count_kmers(bases, offs, k, True)
    for algo in (kmc.ALGO_WALK, kmc.ALGO_STREAM, kmc.ALGO_WALK, kmc.ALGO_SORT):
        with kmc.KmerCounter(k=k, algo=algo) as kc:
            kc.add_batch(bases, offs)
            got = kc.export()
            st = kc.stats()
        if not got.equals(want):
            fails += 1
            w = {(int(h), int(l)): int(c) for h, l, c in zip(want.key_hi, want.key_lo, want.count)}
            g = {(int(h), int(l)): int(c) for h, l, c in zip(got.key_hi, got.key_lo, got.count)}
            missing = [kk for kk in w if kk not in g]; extra = [kk for kk in g if kk not in w]
            diff = [(kk, w[kk], g[kk]) for kk in w if kk in g and g[kk] != w[kk]]
            srt = all((got.key_hi[i], got.key_lo[i]) < (got.key_hi[i+1], got.key_lo[i+1]) for i in range(got.n_distinct - 1))
            print(f"MISMATCH it={it} k={k} algo={names[algo]} used={names[st.algo_last]} bases={int(offs[-1])} distinct {want.n_distinct}/{got.n_distinct} "
                  f"total {want.n_total}/{got.n_total} stat_kmers={st.n_kmers} missing {len(missing)} extra {len(extra)} diff {len(diff)} sorted={srt}", flush=True)
            def show(kk): return kmc.Table(np.array([kk[0]], np.uint64), np.array([kk[1]], np.uint64), np.array([1], np.uint64), k).to_bytes().decode().split("\t")[0]
            for kk in missing[:3]: print("   missing", show(kk), w[kk])
            for kk in extra[:3]: print("   extra  ", show(kk), g[kk])
            for kk, a, b in diff[:3]: print("   count  ", show(kk), a, b)
print("fails", fails, flush=True)
