#!/bin/bash
# the overflow log: parity tests that overflow the LDS memo, then the plateau part of the cardinality sweep
tag=${1:-r03e}
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "second_level or planner or walk or forget_source or wrong_prediction or high_card or baseline_config or sample_fasta or ragged or key_width" > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -8 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/pool_sweep.py --fasta-bytes 1e9 --pools 20,26,32,50,100,300,1000 > gpurun_out/${tag}_pool_sweep.jsonl 2> gpurun_out/${tag}_pool_sweep.err
echo "sweep rc=$?"; tail -2 gpurun_out/${tag}_pool_sweep.err
python3 - <<P
import json
for l in open("gpurun_out/${tag}_pool_sweep.jsonl"):
    d=json.loads(l); print(d["k"], d["pool"], d["algo_last"], d["distinct"], "step_ms", d["step_ms"], "kern_ms", d["count_kernels_ms"], "direct", d["direct_share"], "first", d["first_step_ms"], d["first_step_algo"])
P
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof_pool50 -- python3 tools/pool_sweep.py --fasta-bytes 1e9 --pools 50 --ks 31 --steps 5 > /dev/null 2> gpurun_out/${tag}_prof.err
python3 - <<P
import csv,glob
for f in glob.glob("gpurun_out/${tag}_prof_pool50/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:10]: print("  ", r["Name"][:60].ljust(60), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3), r["Percentage"])
P
