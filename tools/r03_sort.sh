#!/bin/bash
# sort-path round: the tests that go through kmc_extract / kmc_msd, then the all-distinct bench (1 GB, pool 0) at k = 31 and 63
# with a kernel trace each.   usage: tools/r03_sort.sh <tag> [skip-tests]
tag=${1:-r03s}
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
if [ -z "$2" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sample_fasta or key_width or ragged or msd or sort or reference_mode or high_card or wrong_prediction or planner or two_word or finalize or edge_inputs or low_complexity or many_batches or auto_hands or count_file" > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
fi
for k in 31 63; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_k$k -- python3 bench.py --pool 0 --fasta-bytes 1e9 --algo sort --k $k --steps 4 --warmup 2 --no-cpu-baseline --no-read-peak > gpurun_out/${tag}_bench_sort_k$k.json 2> gpurun_out/${tag}_bench_sort_k$k.err
  echo "sort k=$k rc=$?"; tail -2 gpurun_out/${tag}_bench_sort_k$k.err
  python3 - <<P
import json,csv,glob
try:
    d=json.loads([l for l in open("gpurun_out/${tag}_bench_sort_k$k.json") if l.startswith("{")][-1]); print("sort k=$k", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
except Exception as e: print("sort bench unreadable", e)
for f in glob.glob("gpurun_out/${tag}_prof_k$k/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:14]: print(r["Name"][:60].ljust(60), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3), r["Percentage"])
P
done
# un-profiled numbers
for k in 31 63; do
  timeout -k 10 200 python3 bench.py --pool 0 --fasta-bytes 1e9 --algo sort --k $k --steps 5 --warmup 2 --no-cpu-baseline --no-read-peak > gpurun_out/${tag}_bench_sort_plain_k$k.json 2>/dev/null
  python3 - <<P
import json
try:
    d=json.loads([l for l in open("gpurun_out/${tag}_bench_sort_plain_k$k.json") if l.startswith("{")][-1]); print("plain sort k=$k", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
except Exception as e: print("unreadable", e)
P
done
