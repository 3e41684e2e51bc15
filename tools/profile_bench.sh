#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel trace + stats, then PMC passes (each in its own run;
# --pmc is never combined with other trace domains).  Output under gpurun_out/prof_<tag>/.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 5 --warmup 6 --no-cpu-baseline --no-cold $*"
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > $out/trace.log 2>&1
echo "trace rc=$?"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM" \
  "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $out/pmc$i -- python3 bench.py $args > $out/pmc$i.log 2>&1
  echo "pmc$i ($ctrs) rc=$?"
done
find $out -name "*.csv" | head -40
