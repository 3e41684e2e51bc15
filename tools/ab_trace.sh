#!/bin/bash
# kernel timeline of one steady-state step of the k=21 / 1 GB bench in both trees
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
root=$(pwd)
for t in ab_r01 .; do
  name=$(echo $t | tr -d './'); name=${name:-cur}
  (cd $t && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/gpurun_out/abtrace_$name -- python3 bench.py --k 21 --fasta-bytes 1e9 --steps 6 --warmup 4 --no-cpu-baseline --no-cold > /dev/null 2>&1)
  python3 - <<P
import csv,glob
rows=[]
for f in glob.glob("$root/gpurun_out/abtrace_$name/*/*_kernel_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:48]) for r in csv.DictReader(open(f))]
for f in glob.glob("$root/gpurun_out/abtrace_$name/*/*_memory_copy_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")) for r in csv.DictReader(open(f))]
rows.sort()
walk=[i for i,r in enumerate(rows) if "kmc_walk_kernel" in r[2]]
i0=walk[-3]; i1=walk[-2]
t0=rows[i0][0]; prev=t0
print("== $name: one step (from a walk kernel start to the next)")
for s,e,n in rows[i0:i1+1]:
    print(f"{(s-t0)/1e3:8.1f} gap {(s-prev)/1e3:6.1f} dur {(e-s)/1e3:7.1f} {n}")
    prev=e
P
done
