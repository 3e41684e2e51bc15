#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "ragged or every_key_width or sample_fasta or edge or low_complexity or garbage or wrong_prediction" 2>&1 | tail -3
for i in 1 2; do
python bench.py --algo stream --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check --no-read-peak 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('stream k31', d['ms_per_step'], 'kernel', r['kernel_ms'])"
done
python bench.py --algo stream --k 63 --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check --no-read-peak 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('stream k63', d['ms_per_step'], 'kernel', r['kernel_ms'])"
