#!/bin/bash
# copies the lines of tools/r03_evidence_a.sh (gpurun_out/ev3) that are kept to profiles/r03_*
cd $(dirname $0)/..
o=gpurun_out/ev3
last() { grep "^{" "$1" | tail -1; }
last $o/bench_n1.json > profiles/r03_bench_n1.json
last $o/bench_n1_k21.json > profiles/r03_bench_n1_k21.json
last $o/bench_n1_k63.json > profiles/r03_bench_n1_k63.json
last $o/bench_n1_pool0_1GB.json > profiles/r03_bench_n1_pool0_1GB.json
last $o/bench_n1_pool0_1GB_k63.json > profiles/r03_bench_n1_pool0_1GB_k63.json
last $o/bench_n1_stream.json > profiles/r03_bench_n1_stream.json
grep "^{" $o/pool_sweep.jsonl > profiles/r03_pool_sweep.jsonl
last $o/lr_mode.json > profiles/r03_lr_mode.json
cp $o/finalize.txt profiles/r03_finalize_large_tables.txt
last $o/host_path.json > profiles/r03_host_path.json
last $o/file_highcard.json > profiles/r03_file_highcard_1GB.json
cp $o/host_step_times.txt profiles/r03_host_step_times.txt
for n in 2 3; do last $o/rehearsal_n$n.json > profiles/r03_rehearsal_n${n}_one_gpu_gloo.json; done
ls -la profiles/r03_*
