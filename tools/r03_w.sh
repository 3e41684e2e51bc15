#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
for v in main prev main prev; do
  lib=k-mer-count_amd/libkmc.so; [ $v != main ] && lib=k-mer-count_amd/libkmc_$v.so
  KMC_LIB_PATH=$lib python tools/pool_sweep.py --fasta-bytes 1e9 --pools 32,50,100,300 --ks 31 --steps 4 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['k'], d['pool'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'])
"
  KMC_LIB_PATH=$lib python tools/pool_sweep.py --fasta-bytes 1e9 --pools 20,50,100,300 --ks 63 --steps 4 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['k'], d['pool'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'])
"
done
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "second_level or planner or wrong_prediction or high_card or forget_source or walk_two" 2>&1 | tail -2
