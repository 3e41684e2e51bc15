#!/usr/bin/env python3
"""Where the MSD sort's leaf kernel spends its cycles: a DIAGNOSTIC build (tools/build_variant.sh stamps -DKMC_LEAF_STAMPS)
sums thread 0's s_memtime differences per phase over all workgroups.  Shares only -- the stamps' barriers forbid overlaps
the real kernel has.   usage: KMC_LIB_PATH=.../libkmc_stamps.so python tools/leaf_stamps.py [k]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kmc = importlib.import_module("k-mer-count_amd")
import torch
k = int(sys.argv[1]) if len(sys.argv) > 1 else 31
L = kmc.lib()
s = kmc.Synth(seed=2, pool=0)
n, _ = kmc.synth_records_for_bytes(s, int(1e9))
d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda"); d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr()); torch.cuda.synchronize()
out = (C.c_uint64 * 16)()
with kmc.KmerCounter(k=k, algo=kmc.ALGO_SORT) as kc:
    for step in range(3):
        kc.reset(); kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400); kc.finalize()
        L.kmc_debug_leaf_stamps(out, 1)
names = ["load keys", "min/max", "count + prefix", "cursor + scatter", "rank count + rewrite", "large sub-buckets", "run heads", "pairs out"]
v = np.array(list(out)[:8], dtype=np.float64)
print("leaf kernel")
for nm, x in zip(names, v): print(f"  {nm:24s} {x / v.sum() * 100:5.1f} %   {x / 1e6:10.1f} Mcycles")
names = ["tile entry", "digits + rank atomics (+ wait loads)", "prefix", "LDS scatter", "out (stores)"]
v = np.array(list(out)[8:13], dtype=np.float64)
print("scatter kernel")
for nm, x in zip(names, v): print(f"  {nm:38s} {x / v.sum() * 100:5.1f} %   {x / 1e6:10.1f} Mcycles")
