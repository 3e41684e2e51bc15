#!/usr/bin/env python3
"""Time kmc_finalize on tables of n random distinct keys (merge_pairs_device fills the table)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
kmc = importlib.import_module("k-mer-count_amd")
for k in (31, 63):
    for n in ([int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (20_000, 1_000_000, 20_000_000)):
        rng = np.random.default_rng(n + k)
        lo = np.unique(rng.integers(0, 2**62, n, dtype=np.uint64))
        n = lo.size
        hi = rng.integers(0, 2**60, n, dtype=np.uint64) if k > 31 else np.zeros(n, np.uint64)
        cnt = rng.integers(1, 1000, n, dtype=np.uint64)
        perm = rng.permutation(n)
        d_lo = torch.from_numpy(lo[perm].astype(np.int64)).cuda(); d_hi = torch.from_numpy(hi[perm].astype(np.int64)).cuda(); d_c = torch.from_numpy(cnt[perm].astype(np.int64)).cuda()
        with kmc.KmerCounter(k=k) as kc:
            ts = []
            for rep in range(3):
                kc.reset()
                kc.merge_pairs_device(d_hi.data_ptr() if k > 31 else 0, d_lo.data_ptr(), d_c.data_ptr(), n)
                kc.poll()
                torch.cuda.synchronize()
                t0 = time.perf_counter(); nd, nt = kc.finalize(); ts.append(time.perf_counter() - t0)
            assert nd == n and nt == int(cnt.sum())
            print(f"k={k} n={n}: finalize {min(ts)*1e3:.3f} ms (first {ts[0]*1e3:.1f})", flush=True)
