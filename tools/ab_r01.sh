#!/bin/bash
# same-box A/B of the headline bench: round-1 tree (ab_r01/, a git worktree of the round-1 commit) vs the current one
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  for t in ab_r01 .; do
    (cd $t && python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', 'k31 ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])")
  done
done
for t in ab_r01 .; do
  (cd $t && python bench.py --k 21 --fasta-bytes 1e9 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', 'k21 1GB ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])")
  (cd $t && python bench.py --k 63 --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', 'k63 ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])")
done
