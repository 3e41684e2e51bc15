#!/usr/bin/env python3
"""FASTA file -> table through kmc_count_file at several chunk sizes (KMC_INGEST_CHUNK_BYTES) and file sizes."""
import importlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
kmc = importlib.import_module("k-mer-count_amd")
exe = os.path.join(ROOT, "bin", "kmc-genfasta")
for gb in (1, 4):
    with tempfile.NamedTemporaryFile(suffix=".fasta", dir="/dev/shm") as f:
        subprocess.run([exe, "--bytes", str(gb * 10**9), "--seed", "2"], stdout=f, check=True)
        f.flush()
        for mb in (32, 64, 128, 256, 512):
            os.environ["KMC_INGEST_CHUNK_BYTES"] = str(mb << 20)
            ts = []
            for rep in range(3):
                with kmc.KmerCounter(k=31) as kc:
                    t0 = time.perf_counter(); nd, nt = kc.count_file(f.name); ts.append(time.perf_counter() - t0)
            print(json.dumps({"file_GB": gb, "chunk_MiB": mb, "seconds": [round(t, 4) for t in ts], "best_GBps": round(gb / min(ts), 2), "distinct": nd}), flush=True)
