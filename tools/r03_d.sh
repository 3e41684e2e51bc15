#!/bin/bash
# quick sort-path check + the cardinality sweep + LR / finalize measurements
tag=${1:-r03d}
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "msd or sort_path or two_word or reference_mode_matches or small_table or high_card" > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
for k in 31 63; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_k$k -- python3 bench.py --pool 0 --fasta-bytes 1e9 --algo sort --k $k --steps 4 --warmup 2 --no-cpu-baseline --no-read-peak > gpurun_out/${tag}_bench_sort_k$k.json 2> gpurun_out/${tag}_bench_sort_k$k.err
  python3 - <<P
import json,csv,glob
try:
    d=json.loads([l for l in open("gpurun_out/${tag}_bench_sort_k$k.json") if l.startswith("{")][-1]); print("sort k=$k", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
except Exception as e: print("sort bench unreadable", e)
for f in glob.glob("gpurun_out/${tag}_prof_k$k/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:7]: print("  ", r["Name"][:60].ljust(60), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3), r["Percentage"])
P
done
timeout -k 10 500 python tools/pool_sweep.py --fasta-bytes 1e9 > gpurun_out/${tag}_pool_sweep.jsonl 2> gpurun_out/${tag}_pool_sweep.err
echo "sweep rc=$?"; tail -2 gpurun_out/${tag}_pool_sweep.err
python3 - <<P
import json
for l in open("gpurun_out/${tag}_pool_sweep.jsonl"):
    d=json.loads(l); print(d["k"], d["pool"], d["algo_last"], d["distinct"], "step_ms", d["step_ms"], "kern_ms", d["count_kernels_ms"], "direct", d["direct_share"], "first", d["first_step_ms"], d["first_step_algo"])
P
timeout -k 10 200 python3 tools/measure_lr.py > gpurun_out/${tag}_lr.txt 2> gpurun_out/${tag}_lr.err; echo "lr rc=$?"
python3 - <<P
import json
try:
    d=json.loads(open("gpurun_out/${tag}_lr.txt").read().strip().splitlines()[-1])
    for k,v in d.items(): print("lr", k, v["keys"], v["distinct"], "gpu_s", v["gpu_s"], "Gkeys/s", round(v["gpu_keys_per_s"]/1e9,2), v["bit_exact"])
except Exception as e: print("lr unreadable", e)
P
timeout -k 10 200 python3 tools/measure_finalize.py > gpurun_out/${tag}_fin.txt 2> gpurun_out/${tag}_fin.err; echo "fin rc=$?"; cat gpurun_out/${tag}_fin.txt
