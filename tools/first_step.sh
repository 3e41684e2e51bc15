#!/bin/bash
# where does the FIRST step of a fresh ctx on high-cardinality input spend its time?
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
rocprofv3 --kernel-trace --hip-trace --output-format csv -d gpurun_out/first_step -- python3 tools/pool_sweep.py --pools 0 --ks 63 --steps 2 > gpurun_out/first_step.jsonl 2> gpurun_out/first_step.err
cat gpurun_out/first_step.jsonl | cut -c1-300
python3 - <<P
import csv,glob,collections
k=glob.glob("gpurun_out/first_step/*/*_kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(k)), key=lambda r:int(r["Start_Timestamp"]))
# first step = from first walk kernel to first reset after it
iw=[i for i,r in enumerate(rows) if "kmc_walk_kernel" in r["Kernel_Name"]][0]
ir=[i for i,r in enumerate(rows) if i>iw and "kmc_reset_kernel" in r["Kernel_Name"]][0]
t0=int(rows[iw]["Start_Timestamp"]); t1=int(rows[ir]["Start_Timestamp"])
agg=collections.defaultdict(float)
for r in rows[iw:ir]: agg[r["Kernel_Name"].split("(")[0][:44]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
print("first step span ms", (t1-t0)/1e6, "kernel sum ms", sum(agg.values()))
for n,t in sorted(agg.items(), key=lambda kv:-kv[1])[:8]: print("  ", n, round(t,2))
h=glob.glob("gpurun_out/first_step/*/*_hip_api_trace.csv")
if h:
    api=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(h[0])):
        s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
        if s>=t0-1e6 and s<=t1:
            api[r["Function"]][0]+=1; api[r["Function"]][1]+=(e-s)/1e6
    for n,(c,t) in sorted(api.items(), key=lambda kv:-kv[1][1])[:10]: print("  api", n, c, round(t,1))
P
