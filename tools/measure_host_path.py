#!/usr/bin/env python3
"""End-to-end rates that are NOT bench.py's `value` (which times HBM-resident input):
 (1) kmc_add_batch from pageable host buffers (PCIe-inclusive), (2) file -> table through the CLI path
 (host FASTA parse + H2D + kernels).  Prints one JSON line."""
import importlib, json, os, sys, time, tempfile, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kmc = importlib.import_module("k-mer-count_amd")
s = kmc.Synth(seed=2)
n = 2_500_000  # 1 G bases
hb, ho = kmc.synth_reads_host(s, 0, n)
out = {}
with kmc.KmerCounter(k=31) as kc:
    kc.add_batch(hb[:4_000_000], ho[:10_001]); kc.finalize(); kc.reset()   # warm up
    t0 = time.perf_counter(); kc.add_batch(hb, ho); nd, nt = kc.finalize(); dt = time.perf_counter() - t0
    out["host_buffers"] = {"bases": int(hb.size), "seconds": round(dt, 4), "GBps": round(hb.size / dt / 1e9, 2), "kmers_per_s": round(nt / dt, 1)}
exe = os.path.join(ROOT, "bin", "kmc-genfasta")
with tempfile.NamedTemporaryFile(suffix=".fasta", dir="/dev/shm") as f:
    subprocess.run([exe, "--bytes", "1000000000", "--seed", "2"], stdout=f, check=True)
    f.flush()
    with kmc.KmerCounter(k=31) as kc:
        t0 = time.perf_counter(); nd, nt = kc.count_file(f.name); dt = time.perf_counter() - t0
    out["fasta_file_1GB"] = {"seconds": round(dt, 3), "file_GBps": round(1.0 / dt, 3), "kmers_per_s": round(nt / dt, 1), "distinct": nd}
print(json.dumps(out))
