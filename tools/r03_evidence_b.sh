#!/bin/bash
# Round 3, profiled evidence (one gpurun call): kernel trace + stats + PMC passes of the headline command and of the sort path,
# kernel traces of one plateau step, one LR batch and one benchmark step, in-kernel stamps of the diagnostic build.
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
bash tools/profile_bench.sh r03_walk > gpurun_out/prof_r03_walk.log 2>&1; echo "walk prof rc=$?"
bash tools/profile_bench.sh r03_sort --pool 0 --fasta-bytes 1e9 --no-exact-check --no-read-peak > gpurun_out/prof_r03_sort.log 2>&1; echo "sort prof rc=$?"
o=gpurun_out/ev3b; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stream -- python3 bench.py --algo stream --steps 5 --warmup 3 --no-cpu-baseline --no-cold --no-exact-check --no-read-peak > $o/stream.log 2>&1; echo "stream trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/peak -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-cold --no-exact-check > $o/peak.log 2>&1; echo "peak trace rc=$?"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o/step -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-cold --no-exact-check --no-read-peak > /dev/null 2> $o/step.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/p50 -- python3 tools/pool_sweep.py --fasta-bytes 1e9 --pools 50 --ks 31 --steps 4 > /dev/null 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $o/p20k63 -- python3 tools/pool_sweep.py --fasta-bytes 1e9 --pools 20 --ks 63 --steps 4 > /dev/null 2> /dev/null
rocprofv3 --kernel-trace --output-format csv -d $o/lr -- python3 tools/measure_lr.py > /dev/null 2> /dev/null
python3 - <<P
import csv,glob
def rows_of(d):
    rows=[]
    for f in glob.glob(d+"/*/*_kernel_trace.csv"):
        rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:60]) for r in csv.DictReader(open(f))]
    for f in glob.glob(d+"/*/*_memory_copy_trace.csv"):
        rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")) for r in csv.DictReader(open(f))]
    rows.sort(); return rows
def show(title, rows, i0, i1, out):
    t0=rows[i0][0]; prev=t0
    out.write("== "+title+"\n   start us    gap us     dur us  kernel\n")
    for s,e,n in rows[i0:i1]:
        out.write(f"{(s-t0)/1e3:10.1f} {(s-prev)/1e3:9.1f} {(e-s)/1e3:10.1f}  {n}\n"); prev=e
with open("$o/step_timeline.txt","w") as out:
    rows=rows_of("$o/step"); w=[i for i,r in enumerate(rows) if "kmc_walk_kernel" in r[2]]
    show("one benchmark step (10 GB, k=31), from a walk kernel's start to the next one's (rocprofv3 --kernel-trace)", rows, w[-3], w[-2]+1, out)
    rows=rows_of("$o/p50"); w=[i for i,r in enumerate(rows) if "kmc_walk_kernel" in r[2]]
    show("one step of 1 GB, pool 50, k=31 (67,650 distinct 31-mers)", rows, w[-2], w[-1]+1, out)
    rows=rows_of("$o/p20k63"); w=[i for i,r in enumerate(rows) if "kmc_walk_kernel" in r[2]]
    show("one step of 1 GB, pool 20, k=63 (24,080 distinct 63-mers)", rows, w[-2], w[-1]+1, out)
    rows=rows_of("$o/lr"); i0=[i for i,r in enumerate(rows) if "kmc_lr_mer_kernel<0>" in r[2]][-1]
    i1=[i for i,r in enumerate(rows) if "kmc_lr_compose_kernel" in r[2]][-1]
    show("one LR batch (4000 synthetic records, 71 M keys)", rows, i0, i1+1, out)
P
cat $o/step_timeline.txt | grep -v rocclr | head -150
S=k-mer-count_amd/libkmc_stamps.so
{ KMC_LIB_PATH=$S python tools/leaf_stamps.py 31 2>/dev/null | tail -16; KMC_LIB_PATH=$S python tools/leaf_stamps.py 63 2>/dev/null | tail -16; } > $o/sort_stamps.txt; cat $o/sort_stamps.txt
find gpurun_out/prof_r03_walk gpurun_out/prof_r03_sort -name "*.csv" | wc -l
