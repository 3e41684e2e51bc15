#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03m_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03m_tests.log
for i in 1 2 3; do
  python tools/host_step_times.py 2>/dev/null | tail -1
  KMC_NO_MIRROR_SPIN=1 python tools/host_step_times.py 2>/dev/null | tail -1 | sed 's/^/NOSPIN /'
done
python tools/pool_sweep.py --fasta-bytes 1e9 --pools 10,16,20,26,32,50,100,300,1000,0 --ks 31,63 --steps 4 > gpurun_out/r03m_pool_sweep.jsonl 2> gpurun_out/r03m_pool_sweep.err
python3 -c "
import sys,json
for l in open('gpurun_out/r03m_pool_sweep.jsonl'):
    if l.startswith('{'):
        d=json.loads(l); print(d['k'], d['pool'], d['algo_last'], d['distinct'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'], 'direct', d['direct_share'], 'first', d['first_step_ms'], d['first_step_algo'])
"
