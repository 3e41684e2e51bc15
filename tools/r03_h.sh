#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "second_level or planner or msd or two_word" 2>&1 | tail -4
python tools/pool_sweep.py --fasta-bytes 1e9 --pools 26,32,50,100,300,1000 --ks 31,63 --steps 4 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['k'], d['pool'], d['algo_last'], d['distinct'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'], 'direct', d['direct_share'], 'first', d['first_step_ms'], d['first_step_algo'])
"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03h_prof_pool50 -- python3 tools/pool_sweep.py --fasta-bytes 1e9 --pools 50 --ks 31 --steps 5 > /dev/null 2> gpurun_out/r03h_prof.err
python3 - <<P
import csv,glob
for f in glob.glob("gpurun_out/r03h_prof_pool50/*/*_kernel_trace.csv")[:1]:
    rows=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]),r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(f))]
    for n in ("sklog_hist","sklog_partition","sklog_consume","walk_kernel","walk_tail","sk_unfold","small_finalize","compact","leaf"):
        d=[x for x,y in rows if n in y]
        if d: print("   ", n, "max", max(d)/1e3, "us", "n", len(d))
P
for k in 31; do echo "== stamps k=$k"; KMC_LIB_PATH=$(pwd)/k-mer-count_amd/libkmc_stamps.so timeout -k 10 200 python3 tools/leaf_stamps.py $k 2>&1 | tail -16; done
