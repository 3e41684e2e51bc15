#!/bin/bash
# A/B of library variants: headline (k31 10 GB) x3 alternating, k21 1 GB, k63 10 GB.   tools/ab_variants2.sh <name>...
cd ${GRAFT_REPO_ROOT:-.}
run() { v=$1; shift; lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
  KMC_LIB_PATH=$(pwd)/$lib python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$*', 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'), 'frac', d['roofline']['frac'], 'exact', (d['config'].get('exact_full_size_check') or {}).get('bit_exact'))"; }
for i in 1 2 3; do for v in "$@"; do run $v --steps 20 --warmup 5; done; done
for v in "$@"; do run $v --k 21 --fasta-bytes 1e9 --seed 1 --steps 20 --warmup 5; done
for v in "$@"; do run $v --k 63 --steps 10 --warmup 5; done
