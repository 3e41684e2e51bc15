#!/bin/bash
# same-box A/B of library variants on the headline bench: tools/ab_variants.sh <name> [<name> ...]  ("cur" = libkmc.so)
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  for v in "$@"; do
    lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
    KMC_LIB_PATH=$(pwd)/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'k31 ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])"
  done
done
for v in "$@"; do
  lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
  KMC_LIB_PATH=$(pwd)/$lib python bench.py --k 21 --fasta-bytes 1e9 --steps 20 --warmup 5 --no-cpu-baseline --no-exact-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'k21 1GB ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'])"
done
