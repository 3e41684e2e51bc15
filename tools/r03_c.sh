#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
for k in 31 63; do echo "== leaf stamps k=$k"; KMC_LIB_PATH=$(pwd)/k-mer-count_amd/libkmc_stamps.so timeout -k 10 200 python3 tools/leaf_stamps.py $k 2>&1 | tail -9; done
echo "== parity of the global-queue walk variant"
KMC_LIB_PATH=$(pwd)/k-mer-count_amd/libkmc_gq.so timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "walk or sample_fasta or key_width or ragged or baseline_config or garbage or second_level" 2>&1 | tail -4
bash tools/ab_variants.sh cur gq 2>&1
