#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.sh): interleaved bench.py runs, kernel time
# and roofline fraction of each.  usage: tools/ab_bench.sh <out-file> <rounds> <variant>... [-- bench args]
out=$1; rounds=$2; shift 2
vars=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vars+=("$1"); shift; done
[ "$1" == "--" ] && shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
: > $out
for r in $(seq 1 $rounds); do
  for v in "${vars[@]}"; do
    lib=$root/k-mer-count_amd/libkmc_$v.so
    [ "$v" == "main" ] && lib=$root/k-mer-count_amd/libkmc.so
    line=$(KMC_LIB_PATH=$lib python3 bench.py --no-cpu-baseline --steps 20 --warmup 8 "$@" 2>&1 | tail -1)
    echo "$v $line" | python3 -c "
import sys, json
v, rest = sys.stdin.read().split(' ', 1)
try:
    j = json.loads(rest); rf = j['roofline']
    print(v, 'kernel_ms', rf['kernel_ms'], 'frac', rf['frac'], 'ms_per_step', j['ms_per_step'], 'value', j['value'], flush=True)
except Exception as e:
    print(v, 'FAILED', rest[-300:], flush=True)
" | tee -a $out
  done
done
