#!/usr/bin/env python3
"""Stress reproducers for the counting kernels, every result diffed against the CPU oracle.

  stress_walk.py overflow [seed]             KMC_ALGO_WALK on random inputs that overflow its memo tables,
                                             each input four times (races show up as run-to-run differences)
  stress_walk.py fresh [seed] [iters]        same-size batches with DIFFERENT data every time through
                                             walk / stream / walk / sort (freed device buffers get reused:
                                             exposes stale-cache reads as well as races)
  stress_walk.py replay <seed> <iter> <reps> replays ONE input of `fresh` many times through the walk kernel
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
kmc = importlib.import_module("k-mer-count_amd")
import oracle_py

ACGT = np.frombuffer(b"ACGT", np.uint8)


def report(tag, k, got, want):
    w = {(int(h), int(l)): int(c) for h, l, c in zip(want.key_hi, want.key_lo, want.count)}
    g = {(int(h), int(l)): int(c) for h, l, c in zip(got.key_hi, got.key_lo, got.count)}
    missing = [(kk, w[kk]) for kk in w if kk not in g]
    extra = [(kk, g[kk]) for kk in g if kk not in w]
    diff = [(kk, w[kk], g[kk]) for kk in w if kk in g and g[kk] != w[kk]]
    print(f"MISMATCH {tag} k={k} distinct {want.n_distinct} got {got.n_distinct} total {want.n_total} got {got.n_total}: "
          f"missing {len(missing)} extra {len(extra)} diffcount {len(diff)}", flush=True)
    one = lambda kk, c: kmc.Table(np.array([kk[0]], np.uint64), np.array([kk[1]], np.uint64), np.array([c], np.uint64), k).to_bytes().decode().strip()
    for kk, c in missing[:3]:
        print("   missing", one(kk, c))
    for kk, c in extra[:3]:
        print("   extra  ", one(kk, c))
    for kk, a, b in diff[:3]:
        print("   count  ", one(kk, a), "got", b)


def fresh_input(rng, it):
    k = int(rng.choice([17, 24, 31, 47]))
    lens = rng.integers(0, 301, 300) if it % 2 else np.full(300, 150)
    offs = np.zeros(301, np.uint64); offs[1:] = np.cumsum(lens)
    return k, ACGT[rng.integers(0, 4, int(offs[-1]))].copy(), offs


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "overflow"
    fails = 0
    if mode == "overflow":
        rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 124)
        for trial in range(40):
            k = int(rng.choice([5, 16, 24, 31]))
            nreads = int(rng.choice([50, 300, 3000]))
            lens = rng.integers(0, int(rng.choice([60, 300, 416])) + 1, nreads)
            offs = np.zeros(nreads + 1, np.uint64); offs[1:] = np.cumsum(lens)
            bases = ACGT[rng.integers(0, 4, int(offs[-1]))].copy()
            want = oracle_py.count_kmers(bases, offs, k, True)
            for rep in range(4):
                with kmc.KmerCounter(k=k, algo=kmc.ALGO_WALK) as kc:
                    kc.add_batch(bases, offs)
                    got = kc.export()
                if not got.equals(want):
                    fails += 1
                    report(f"trial {trial} rep {rep} reads={nreads}", k, got, want)
    elif mode == "fresh":
        rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
        iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
        for it in range(iters):
            k, bases, offs = fresh_input(rng, it)
            want = oracle_py.count_kmers(bases, offs, k, True)
            for algo in (kmc.ALGO_WALK, kmc.ALGO_STREAM, kmc.ALGO_WALK, kmc.ALGO_SORT):
                with kmc.KmerCounter(k=k, algo=algo) as kc:
                    kc.add_batch(bases, offs)
                    got = kc.export()
                if not got.equals(want):
                    fails += 1
                    report(f"iter {it} algo {algo}", k, got, want)
    elif mode == "replay":
        seed, target, reps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        rng = np.random.default_rng(seed)
        for it in range(target + 1):
            k, bases, offs = fresh_input(rng, it)
        want = oracle_py.count_kmers(bases, offs, k, True)
        for rep in range(reps):
            with kmc.KmerCounter(k=k, algo=kmc.ALGO_WALK) as kc:
                kc.add_batch(bases, offs)
                got = kc.export()
            if not got.equals(want):
                fails += 1
                report(f"replay {rep}", k, got, want)
    else:
        raise SystemExit(__doc__)
    print("fails", fails, flush=True)


if __name__ == "__main__":
    main()
