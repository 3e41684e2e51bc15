#!/bin/bash
# same-box A/B of library variants on the middle of the cardinality sweep + the headline.  tools/ab_sweep.sh <name>...
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
  KMC_LIB_PATH=$(pwd)/$lib python tools/pool_sweep.py --fasta-bytes 1e9 --pools 20,26,32,50,100,1000 --ks 31,63 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$v', d['k'], d['pool'], d['algo_last'], 'step', d['step_ms'], 'kern', d['count_kernels_ms'])"
done
for i in 1 2; do for v in "$@"; do
  lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
  KMC_LIB_PATH=$(pwd)/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'k31 10GB ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), d['config']['exact_full_size_check']['bit_exact'])"
done; done
