#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "finalize or merge or partition or planner or empty_and_short or count_file or rccl or slab" 2>&1 | tail -3
python tools/measure_finalize.py 70000,300000,1000000 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03v_fin -- python3 tools/measure_finalize.py 70000,1000000 > /dev/null 2>&1
python3 - <<P
import csv,glob
for f in glob.glob("gpurun_out/r03v_fin/*/*_kernel_stats.csv")[:1]:
    for r in list(csv.DictReader(open(f)))[:14]: print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), "avg", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1))
P
bash tools/r03_stress.sh 777
