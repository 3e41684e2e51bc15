#!/bin/bash
# kernel trace of one point of the cardinality sweep.   usage: bash tools/prof_pool.sh <tag> <pool> <k> [algo]
tag=$1; pool=$2; k=$3; algo=${4:-auto}
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python3 tools/pool_sweep.py --pools $pool --ks $k --steps 3 --algo $algo > gpurun_out/${tag}.jsonl 2> gpurun_out/${tag}.err
cat gpurun_out/${tag}.jsonl
python3 - <<P
import csv,glob
for f in glob.glob("gpurun_out/prof_${tag}/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:16]: print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), round(float(r["AverageNs"])/1e6,3), round(float(r["TotalDurationNs"])/1e6,1), r["Percentage"])
P
