import importlib, os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import torch
kmc = importlib.import_module("k-mer-count_amd")
s = kmc.Synth(seed=2)
for ntiles_per_wave in (1, 2, 4, 8):
    n = 4096 * 64 * ntiles_per_wave
    d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda"); d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr()); torch.cuda.synchronize()
    kc = kmc.KmerCounter(k=31)
    ts = []
    for i in range(12):
        kc.reset(); kc.add_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, n * 400, 400)
        try: kc.finalize()
        except Exception as e: pass
        ts.append(kc.stats().kernel_ms_last)
    print("tiles/wave", ntiles_per_wave, "kernel_us", [round(t * 1e3, 1) for t in ts[6:]], flush=True)
    kc.close()
