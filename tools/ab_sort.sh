#!/bin/bash
# same-box A/B of library variants on the sort path (1 GB pool 0, k=31 and k=63) and the LR mode.  tools/ab_sort.sh <name>...
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do for v in "$@"; do
  lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
  for k in 31 63; do
  KMC_LIB_PATH=$(pwd)/$lib python bench.py --pool 0 --fasta-bytes 1e9 --algo sort --k $k --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'k$k pool0 sort ms_per_step', d['ms_per_step'])"
  done
done; done
for v in "$@"; do
  lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
  KMC_LIB_PATH=$(pwd)/$lib python tools/measure_lr.py 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print('$v', 'lr', k, v['gpu_kernel_ms'], v['bit_exact'])"
done
