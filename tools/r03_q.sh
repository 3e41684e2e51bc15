#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "counted_spans" 2>&1 | tail -3
for pk in "50 31" "100 31" "100 63"; do set -- $pk
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r03q_p$1_k$2 -- python3 tools/pool_sweep.py --fasta-bytes 1e9 --pools $1 --ks $2 --steps 4 > /dev/null 2> /dev/null
python3 - <<P
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/r03q_p$1_k$2/*/*_kernel_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:46]) for r in csv.DictReader(open(f))]
for f in glob.glob("gpurun_out/r03q_p$1_k$2/*/*_memory_copy_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")) for r in csv.DictReader(open(f))]
rows.sort()
walk=[i for i,r in enumerate(rows) if "kmc_walk_kernel" in r[2]]
i0=walk[-2]; i1=walk[-1]; t0=rows[i0][0]; prev=t0
print("== pool $1 k=$2 step")
for s,e,n in rows[i0:i1+1]:
    print(f"{(s-t0)/1e3:9.1f} gap {(s-prev)/1e3:7.1f} dur {(e-s)/1e3:8.1f} {n}")
    prev=e
P
done
