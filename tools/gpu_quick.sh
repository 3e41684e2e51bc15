#!/bin/bash
# quick GPU check of the sort machinery: the tests that exercise it, then kernel traces of the clustered-key cases and LR
tag=$1
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "msd or lr or LR or sort or finalize or cardinality or recovered or golden or kat or baseline" > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/prof_pool.sh ${tag}_63_1000 1000 63 && bash tools/prof_pool.sh ${tag}_31_0 0 31 && bash tools/prof_pool.sh ${tag}_63_0 0 63 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_lr -- python3 tools/measure_lr.py > gpurun_out/${tag}_lr.txt 2> gpurun_out/${tag}_lr.err; echo "lr rc=$?"
python3 - <<P
import json,csv,glob
try:
    d=json.loads(open("gpurun_out/${tag}_lr.txt").read().strip().splitlines()[-1])
    for k,v in d.items(): print("lr", k, v["keys"], v["distinct"], "gpu_s", v["gpu_s"], "kern_ms", v["gpu_kernel_ms"], "Gkeys/s", round(v["gpu_keys_per_s"]/1e9,2), v["bit_exact"])
except Exception as e: print("lr unreadable", e)
for f in glob.glob("gpurun_out/prof_${tag}_lr/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:12]: print(r["Name"][:52].ljust(52), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e6,3), r["Percentage"])
P
python3 tools/measure_lr.py | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print('lr unprofiled', k, 'gpu_s', v['gpu_s'], 'kern_ms', v['gpu_kernel_ms'], 'Gkeys/s', round(v['gpu_keys_per_s']/1e9,2), v['bit_exact'])"
