#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "lr or LR or sort or merge or finalize or stress or reference_mode or golden" 2>&1 | tail -3
python tools/measure_lr.py 2>/dev/null | tail -1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r03p_lr -- python3 tools/measure_lr.py > /dev/null 2> gpurun_out/r03p_lr.err
python3 - <<P
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/r03p_lr/*/*_kernel_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:60]) for r in csv.DictReader(open(f))]
for f in glob.glob("gpurun_out/r03p_lr/*/*_memory_copy_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")) for r in csv.DictReader(open(f))]
rows.sort()
i0=[i for i,r in enumerate(rows) if "kmc_lr_mer_kernel<0>" in r[2]][-1]
t0=rows[i0][0]; prev=t0
print("== last LR batch")
for s,e,n in rows[i0:]:
    if (s-t0)/1e3 > 6000: break
    print(f"{(s-t0)/1e3:9.1f} gap {(s-prev)/1e3:7.1f} dur {(e-s)/1e3:8.1f} {n}")
    prev=e
P
python tools/pool_sweep.py --fasta-bytes 1e9 --pools 0,1000 --ks 31 --steps 3 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['k'], d['pool'], d['algo_last'], d['distinct'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'])
"
