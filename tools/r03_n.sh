#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
S=k-mer-count_amd/libkmc_stamps.so
KMC_LIB_PATH=$S python tools/walk_stamps.py 1e9 21 2>/dev/null | tail -14
KMC_LIB_PATH=$S python tools/walk_stamps.py 1e9 31 2>/dev/null | tail -14
KMC_LIB_PATH=$S python tools/walk_stamps.py 10e9 31 2>/dev/null | tail -14
python tools/measure_lr.py 2>/dev/null | tail -1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r03n_lr -- python3 tools/measure_lr.py > /dev/null 2> gpurun_out/r03n_lr.err
python3 - <<P
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/r03n_lr/*/*_kernel_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:60]) for r in csv.DictReader(open(f))]
for f in glob.glob("gpurun_out/r03n_lr/*/*_memory_copy_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")) for r in csv.DictReader(open(f))]
rows.sort()
# the last LR batch: from the last kmc_lr_mer_kernel<0> to the end
i0=[i for i,r in enumerate(rows) if "kmc_lr_mer_kernel<0>" in r[2]][-1]
t0=rows[i0][0]; prev=t0
print("== last LR batch")
for s,e,n in rows[i0:]:
    print(f"{(s-t0)/1e3:9.1f} gap {(s-prev)/1e3:7.1f} dur {(e-s)/1e3:8.1f} {n}")
    prev=e
P
