"""Diagnostic: per-wave phase stamps of kmc_walk_kernel.  Build the variant with tools/build_walk_stamps.py, then
KMC_LIB_PATH=k-mer-count_amd/libkmc_wstamps.so python tools/walk_stamps.py"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
kmc = importlib.import_module("k-mer-count_amd")
s = kmc.Synth(seed=2)
n, _ = kmc.synth_records_for_bytes(s, int(float(sys.argv[1]) if len(sys.argv) > 1 else 10e9))
d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda"); d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr()); torch.cuda.synchronize()
kc = kmc.KmerCounter(k=31)
L = kmc.lib()
buf = np.zeros(256 * 16 * 8, np.uint64)
for it in range(8):
    kc.reset(); kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400); kc.finalize()
    assert L.kmc_debug_walk_stamps(C.c_void_p(buf.ctypes.data)) == 0
    if it < 5: continue
    st = buf.reshape(256, 16, 8).astype(np.int64)
    t0 = st[:, :, 0].min()
    us = lambda a: (a - t0) / 100.0   # 100 MHz ticks -> us
    def desc(name, a):
        a = us(a); print(f"  {name:28s} min {a.min():8.1f}  p50 {np.median(a):8.1f}  max {a.max():8.1f} us")
    print(f"iter {it}: kernel_ms_last {kc.stats().kernel_ms_last:.4f}")
    desc("wave start", st[:, :, 0]); desc("LDS init done", st[:, :, 1]); desc("first tile loaded", st[:, :, 2]); desc("first tile stepped", st[:, :, 3])
    desc("tile loop done", st[:, :, 4]); desc("wave end (after flush)", st[:, :, 5])
    wg_done = st[:, :, 4].max(axis=1); desc("per-WG last wave loop done", wg_done)
    wd = us(wg_done)
    print("  per-WG loop done by blockIdx % 8 (XCD): " + " ".join(f"{wd[x::8].mean():7.1f}" for x in range(8)) + f"   (std within {np.mean([wd[x::8].std() for x in range(8)]):.1f} us)")
    print("  per-WG loop done by blockIdx // 32:     " + " ".join(f"{wd[32 * x:32 * x + 32].mean():7.1f}" for x in range(8)))
    print("  tiles per wave:", np.unique(st[:, :, 6], return_counts=True))
    dur = us(st[:, :, 4]) - us(st[:, :, 1])
    for t in np.unique(st[:, :, 6])[[0, -1]]:
        m = st[:, :, 6] == t
        if m.any(): print(f"  loop time of waves with {t} tiles: min {dur[m].min():.1f} p50 {np.median(dur[m]):.1f} max {dur[m].max():.1f} us")
