#!/usr/bin/env python3
"""Cold-vs-warm memo and per-phase host overhead of one bench step (same box, same process)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
kmc = importlib.import_module("k-mer-count_amd")
s = kmc.Synth(seed=2)
n, _ = kmc.synth_records_for_bytes(s, int(10e9))
d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda")
d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr())
torch.cuda.synchronize()
for k in (31, 63):
    kc = kmc.KmerCounter(k=k)
    out = []
    for i in range(8):
        t0 = time.perf_counter(); kc.reset(); torch.cuda.synchronize(); t1 = time.perf_counter()
        kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400); t2 = time.perf_counter()
        kc.finalize(); t3 = time.perf_counter()
        out.append((round(kc.stats().kernel_ms_last, 4), round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3), round((t3 - t2) * 1e3, 3)))
    print("k", k, "per step (kernel_ms, reset_ms, launch_ms, finalize_ms incl. kernel):", out, flush=True)
    kc.close()
