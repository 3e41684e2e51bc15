#!/usr/bin/env python3
"""Replays ONE input of stress_walk3 (seed, iteration) many times through KMC_ALGO_WALK and
locates the corrupted base of every mismatch."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
kmc = importlib.import_module("k-mer-count_amd")
import oracle_py
seed, target, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for it in range(target + 1):
    k = int(rng.choice([17, 24, 31, 47]))
    nreads = 300
    lens = rng.integers(0, 301, nreads) if it % 2 else np.full(nreads, 150)
    offs = np.zeros(nreads + 1, np.uint64); offs[1:] = np.cumsum(lens)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(offs[-1]))].copy()
print("input: k", k, "bases", int(offs[-1]), flush=True)
want_f = oracle_py.count_kmers(bases, offs, k, False)   # forward: easier to locate
fails = 0
seq = bases.tobytes()
for rep in range(reps):
    with kmc.KmerCounter(k=k, canonical=False, algo=kmc.ALGO_WALK) as kc:
        kc.add_batch(bases, offs)
        got = kc.export()
    if not got.equals(want_f):
        fails += 1
        w = {(int(h), int(l)) for h, l in zip(want_f.key_hi, want_f.key_lo)}
        g = {(int(h), int(l)): int(c) for h, l, c in zip(got.key_hi, got.key_lo, got.count)}
        missing = [kk for kk in w if kk not in g]
        def txt(kk): return kmc.Table(np.array([kk[0]], np.uint64), np.array([kk[1]], np.uint64), np.array([1], np.uint64), k).to_bytes().decode().split("\t")[0]
        locs = sorted(seq.find(txt(kk).encode()) for kk in missing)
        extra = [txt(kk) for kk in g if kk not in w]
        rd = [int(np.searchsorted(offs, p, side="right") - 1) for p in locs]
        print(f"rep {rep}: missing {len(missing)} at stream positions {locs[:8]} reads {sorted(set(rd))} read_start {[int(offs[r]) for r in sorted(set(rd))]} extra0 {extra[:1]} miss0 {txt(missing[0]) if missing else None}", flush=True)
print("fails", fails, "of", reps, flush=True)
