#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
  export KMC_LIB_PATH=$(pwd)/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/absk_$v -- python3 tools/pool_sweep.py --fasta-bytes 1e9 --pools 50 --ks 31 --steps 4 > gpurun_out/absk_$v.json 2> gpurun_out/absk_$v.err
  python3 - <<P
import csv,glob
print("== $v")
for f in glob.glob("gpurun_out/absk_$v/*/*_kernel_trace.csv")[:1]:
    rows=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]),r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(f))]
    for n in ("sklog_partition","sklog_consume","walk_kernel"):
        d=[x for x,y in rows if n in y]
        if d: print("   ", n, "max", max(d)/1e3, "us")
P
done
