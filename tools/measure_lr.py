#!/usr/bin/env python3
"""Reference mode (27+gap+27, sizes 80..=140) throughput: GPU path vs the CPU oracle."""
import importlib, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
kmc = importlib.import_module("k-mer-count_amd")
import oracle_py
out = {}
for name, (bases, offs) in {"sample.fasta": kmc.parse_fasta(os.path.join(ROOT, "tests/golden/sample.fasta")),
                            "synthetic 4000 records": kmc.synth_reads_host(kmc.Synth(seed=2), 0, 4000)}.items():
    t0 = time.perf_counter(); want = oracle_py.count_lr(bases, offs); t_cpu = time.perf_counter() - t0
    with kmc.KmerCounter(mode=kmc.MODE_LR) as kc:
        kc.add_batch(bases, offs); kc.finalize(); kc.reset()          # warm-up (table growth)
        t0 = time.perf_counter(); kc.add_batch(bases, offs); nd, nt = kc.finalize(); t_gpu = time.perf_counter() - t0
        k_ms = kc.stats().kernel_ms_last
        ok = kc.export().equals(want)
    out[name] = {"bases": int(bases.size), "keys": int(nt), "distinct": int(nd), "gpu_s": round(t_gpu, 4), "gpu_kernel_ms": round(k_ms, 3),
                 "cpu_oracle_s": round(t_cpu, 3), "gpu_keys_per_s": round(nt / t_gpu), "bit_exact": bool(ok)}
print(json.dumps(out))
