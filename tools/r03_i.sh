#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "small_table or second_level or msd or finalize_async or merge_and" 2>&1 | tail -4
for k in 31 63; do
  timeout -k 10 200 python3 bench.py --pool 0 --fasta-bytes 1e9 --algo sort --k $k --steps 5 --warmup 2 --no-cpu-baseline --no-read-peak 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('plain sort k=$k', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
done
python tools/pool_sweep.py --fasta-bytes 1e9 --pools 26,32,50,100,300 --ks 31,63 --steps 4 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['k'], d['pool'], d['algo_last'], d['distinct'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'], 'direct', d['direct_share'], 'first', d['first_step_ms'], d['first_step_algo'])
"
for a in walk sort auto; do python tools/pool_sweep.py --fasta-bytes 1e9 --pools 1000 --ks 31,63 --steps 4 --algo $a 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$a', d['k'], d['pool'], d['algo_last'], d['distinct'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'], 'direct', d['direct_share'], 'first', d['first_step_ms'], d['first_step_algo'])
"; done
