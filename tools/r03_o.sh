#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "walk or garbage or long_read or ragged or empty or stress" 2>&1 | tail -3
bash tools/ab_bench.sh gpurun_out/r03o_ab_1g_k21.txt 3 main prev t0 s2 t16 t4s8 -- --fasta-bytes 1e9 --k 21 --no-cold --no-exact-check --no-read-peak > /dev/null 2>&1
cat gpurun_out/r03o_ab_1g_k21.txt
bash tools/ab_bench.sh gpurun_out/r03o_ab_10g_k31.txt 2 main prev t0 s2 t16 t4s8 -- --no-cold --no-exact-check --no-read-peak > /dev/null 2>&1
cat gpurun_out/r03o_ab_10g_k31.txt
KMC_LIB_PATH=k-mer-count_amd/libkmc_stamps.so python tools/walk_stamps.py 1e9 21 2>/dev/null | tail -14
