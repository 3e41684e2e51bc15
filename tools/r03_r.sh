#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03r_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03r_tests.log
python tools/pool_sweep.py --fasta-bytes 1e9 --pools 32,50,100,300 --ks 31,63 --steps 4 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['k'], d['pool'], d['algo_last'], d['distinct'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'])
"
