#!/bin/bash
# One GPU round of the build loop: full GPU test suite, then the benches and the sort-path kernel
# profile.  Output under gpurun_out/ with the given tag.   usage: bash tools/gpu_round.sh <tag> [tests-filter]
tag=$1; filt=${2:-}
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
if [ -n "$filt" ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -x -q -k "$filt" > gpurun_out/${tag}_tests.log 2>&1
else
  timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
fi
rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 100 python bench.py --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_walk.json 2> gpurun_out/${tag}_bench_walk.err
echo "walk rc=$?"; python3 - <<P
import json
try:
    d=json.load(open("gpurun_out/${tag}_bench_walk.json")); print("walk", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
except Exception as e: print("walk bench unreadable", e)
P
timeout -k 10 200 python bench.py --algo stream --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_stream.json 2> gpurun_out/${tag}_bench_stream.err
echo "stream rc=$?"; python3 - <<P
import json
try:
    d=json.load(open("gpurun_out/${tag}_bench_stream.json")); print("stream", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
except Exception as e: print("stream bench unreadable", e)
P
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_sort -- python3 bench.py --pool 0 --fasta-bytes 1e9 --algo sort --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_sort.json 2> gpurun_out/${tag}_bench_sort.err
echo "sort rc=$?"; tail -2 gpurun_out/${tag}_bench_sort.err
python3 - <<P
import json,csv,glob
try:
    d=json.load(open("gpurun_out/${tag}_bench_sort.json")); print("sort", d["value"], d["ms_per_step"])
except Exception as e: print("sort bench unreadable", e)
for f in glob.glob("gpurun_out/prof_${tag}_sort/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:18]: print(r["Name"][:52].ljust(52), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3), r["Percentage"])
P

timeout -k 10 400 python tools/pool_sweep.py --fasta-bytes 1e9 > gpurun_out/${tag}_pool_sweep.jsonl 2> gpurun_out/${tag}_pool_sweep.err
echo "sweep rc=$?"; tail -2 gpurun_out/${tag}_pool_sweep.err
python3 - <<P
import json
for l in open("gpurun_out/${tag}_pool_sweep.jsonl"):
    d=json.loads(l); print(d["k"], d["pool"], d["algo_last"], d["distinct"], "step_ms", d["step_ms"], "kern_ms", d["count_kernels_ms"], "direct", d["direct_share"], "first", d["first_step_ms"])
P
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_lr -- python3 tools/measure_lr.py > gpurun_out/${tag}_lr.txt 2> gpurun_out/${tag}_lr.err; echo "lr rc=$?"
python3 - <<P
import json,csv,glob
try:
    d=json.loads(open("gpurun_out/${tag}_lr.txt").read().strip().splitlines()[-1])
    for k,v in d.items(): print("lr", k, v["keys"], v["distinct"], "gpu_s", v["gpu_s"], "Gkeys/s", round(v["gpu_keys_per_s"]/1e9,2), v["bit_exact"])
except Exception as e: print("lr unreadable", e)
for f in glob.glob("gpurun_out/prof_${tag}_lr/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:10]: print(r["Name"][:52].ljust(52), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3), r["Percentage"])
P
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_fin -- python3 tools/measure_finalize.py > gpurun_out/${tag}_fin.txt 2> gpurun_out/${tag}_fin.err; echo "fin rc=$?"; cat gpurun_out/${tag}_fin.txt
python3 - <<P
import csv,glob
for f in glob.glob("gpurun_out/prof_${tag}_fin/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:52].ljust(52), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3), r["Percentage"])
P
