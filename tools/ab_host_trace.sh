#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
root=$(pwd)
for t in ab_r01 .; do
  name=$(echo $t | tr -d './'); name=${name:-cur}
  (cd $t && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $root/gpurun_out/hosttrace_$name -- python3 tools/measure_host_path.py > /dev/null 2>&1)
  python3 - <<P
import csv,glob,collections
rows=[]
for f in glob.glob("$root/gpurun_out/hosttrace_$name/*/*_kernel_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(f))]
for f in glob.glob("$root/gpurun_out/hosttrace_$name/*/*_memory_copy_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")) for r in csv.DictReader(open(f))]
rows.sort()
# the file->table phase = after the last big gap (genfasta subprocess): take the last 400 ms of activity
tend=rows[-1][1]
sel=[r for r in rows if r[0] > tend-200e6]
agg=collections.defaultdict(lambda:[0,0.0])
for s,e,n in sel: agg[n][0]+=1; agg[n][1]+=(e-s)/1e6
print("== $name: last 200 ms of device activity; span", (sel[-1][1]-sel[0][0])/1e6, "ms")
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:12]: print("  ", n.ljust(46), c, round(t,3))
walks=[(s,e) for s,e,n in sel if "kmc_walk_kernel" in n]
print("   walk launches:", len(walks), "first at", (walks[0][0]-sel[0][0])/1e6 if walks else None)
prev=None
for s,e,n in sel:
    if "kmc_walk_kernel" in n:
        print(f"     walk at {(s-sel[0][0])/1e6:8.3f} ms dur {(e-s)/1e6:6.3f}")
P
done
