#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/c2_tests.log 2>&1; rc=$?; tail -4 gpurun_out/c2_tests.log; [ $rc -ne 0 ] && exit $rc
python tools/measure_host_path.py 2>/dev/null | tail -1 | cut -c1-300
python tools/pool_sweep.py --fasta-bytes 1e9 --pools 10,26,32,50,100,1000 --ks 31,63 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['k'], d['pool'], d['algo_last'], d['step_ms'], d['count_kernels_ms'], 'first', d['first_step_ms'])"
for v in cur bigtile cur bigtile; do
  lib=k-mer-count_amd/libkmc_$v.so; [ "$v" = cur ] && lib=k-mer-count_amd/libkmc.so
  KMC_LIB_PATH=$(pwd)/$lib python bench.py --pool 0 --fasta-bytes 1e9 --algo sort --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'pool0 sort ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))"
done
