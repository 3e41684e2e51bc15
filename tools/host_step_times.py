#!/usr/bin/env python3
"""Host time per call of the benchmark step (reset, add_batch_device, finalize) in its steady state, and the step time:
where the microseconds between a step's synchronisation and the next step's first launch go.   usage: [fasta_bytes]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
kmc = importlib.import_module("k-mer-count_amd")
fb = float(sys.argv[1]) if len(sys.argv) > 1 else 10e9
s = kmc.Synth(seed=2, pool=10)
n, _ = kmc.synth_records_for_bytes(s, int(fb))
d_b = torch.empty(n * 400 + 64, dtype=torch.uint8, device="cuda"); d_o = torch.empty(n + 1, dtype=torch.int64, device="cuda")
kmc.synth_reads_device(s, 0, n, d_b.data_ptr(), d_o.data_ptr()); torch.cuda.synchronize()
pb, po = d_b.data_ptr(), d_o.data_ptr()
with kmc.KmerCounter(k=31) as kc:
    for _ in range(8):
        kc.reset(); kc.add_batch_device(pb, po, n, n * 400, 400); kc.finalize()
    t = {"reset": 0, "add": 0, "finalize": 0}
    K = 40
    t0 = time.perf_counter_ns()
    for _ in range(K):
        a = time.perf_counter_ns(); kc.reset()
        b = time.perf_counter_ns(); kc.add_batch_device(pb, po, n, n * 400, 400)
        c = time.perf_counter_ns(); kc.finalize()
        d = time.perf_counter_ns()
        t["reset"] += b - a; t["add"] += c - b; t["finalize"] += d - c
    tot = time.perf_counter_ns() - t0
    st = kc.stats()
    print({k: round(v / K / 1e3, 2) for k, v in t.items()}, "us per call;", "step", round(tot / K / 1e3, 1), "us; kernel", round(st.kernel_ms_last * 1e3, 1), "us")
