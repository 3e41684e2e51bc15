#!/usr/bin/env python3
"""Where a walk launch's time goes at a given batch size: a DIAGNOSTIC build (tools/build_variant.sh stamps -DKMC_WALK_STAMPS)
records the 100 MHz clock at the kernel's milestones per workgroup; printed relative to the first workgroup's entry.
   usage: KMC_LIB_PATH=.../libkmc_stamps.so python tools/walk_stamps.py [fasta_bytes] [k]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kmc = importlib.import_module("k-mer-count_amd")
import torch
fb = float(sys.argv[1]) if len(sys.argv) > 1 else 1e9
k = int(sys.argv[2]) if len(sys.argv) > 2 else 21
L = kmc.lib()
s = kmc.Synth(seed=2, pool=10)
n, _ = kmc.synth_records_for_bytes(s, int(fb))
d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda"); d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr()); torch.cuda.synchronize()
out = (C.c_uint64 * (256 * 24))()
with kmc.KmerCounter(k=k) as kc:
    for step in range(6):
        kc.reset(); kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400); kc.finalize()
    st = kc.stats()
    L.kmc_debug_walk_stamps(out, 256 * 24)
v = np.array(list(out), dtype=np.float64).reshape(256, 24) / 100.0   # us
t0 = v[:, 0].min()
v -= t0
def row(name, x): print(f"  {name:34s} min {x.min():8.1f}  median {np.median(x):8.1f}  max {x.max():8.1f} us")
print(f"walk kernel, {fb / 1e9:g} GB, k={k}: kernel_ms_last {st.kernel_ms_last:.4f}")
row("entry", v[:, 0]); row("LDS initialised (since entry)", v[:, 1] - v[:, 0])
row("wave 0 stepped its first tile", v[:, 21])
w = v[:, 2:18]
row("first wave out of the tile loop", w.min(axis=1)); row("last wave out of the tile loop", w.max(axis=1))
row("  spread within a workgroup", w.max(axis=1) - w.min(axis=1))
row("dense flush (barrier .. list)", v[:, 19] - v[:, 18]); row("items + memo save", v[:, 20] - v[:, 19]); row("end", v[:, 20])
x = v[:, 20].reshape(8, 32) if False else None
byx = [v[i::8, 20].max() for i in range(8)]
print("  end by XCD (workgroup % 8):", " ".join(f"{b:.1f}" for b in byx))
