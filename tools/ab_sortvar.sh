#!/bin/bash
# same-box A/B of library variants on the all-distinct sort bench (1 GB, pool 0): kernel stats per variant
# usage: tools/ab_sortvar.sh <k> <name> [<name> ...]   ("cur" = libkmc.so)
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
k=$1; shift
for v in "$@"; do
  lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
  export KMC_LIB_PATH=$(pwd)/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abs_${v}_k$k -- python3 bench.py --pool 0 --fasta-bytes 1e9 --algo sort --k $k --steps 4 --warmup 2 --no-cpu-baseline --no-read-peak > gpurun_out/abs_${v}_k$k.json 2> gpurun_out/abs_${v}_k$k.err
  python3 - <<P
import json,csv,glob
try:
    d=json.loads([l for l in open("gpurun_out/abs_${v}_k$k.json") if l.startswith("{")][-1]); print("== $v k=$k ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
except Exception as e: print("$v unreadable", e)
for f in glob.glob("gpurun_out/abs_${v}_k$k/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:5]: print("   ", r["Name"][:56].ljust(56), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3))
P
done
