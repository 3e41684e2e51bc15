#!/bin/bash
# the round's un-profiled evidence runs; output under gpurun_out/final_<tag>/
tag=$1
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/final_$tag; mkdir -p $o
python bench.py > $o/bench_n1.json 2> $o/bench_n1.err; echo "n1 rc=$?"
python bench.py --k 21 --fasta-bytes 1e9 --seed 1 > $o/bench_n1_k21.json 2> $o/bench_k21.err; echo "k21 rc=$?"
python bench.py --k 63 > $o/bench_n1_k63.json 2> $o/bench_k63.err; echo "k63 rc=$?"
python bench.py --pool 0 --fasta-bytes 1e9 --steps 5 --warmup 2 > $o/bench_n1_pool0_1GB.json 2> $o/bench_pool0.err; echo "pool0 rc=$?"
python bench.py --algo stream --steps 5 --warmup 2 --no-cpu-baseline > $o/bench_n1_stream.json 2> $o/bench_stream.err; echo "stream rc=$?"
python tools/pool_sweep.py --fasta-bytes 1e9 > $o/pool_sweep.jsonl 2> $o/pool_sweep.err; echo "sweep rc=$?"
python tools/measure_lr.py > $o/lr_mode.txt 2> $o/lr.err; echo "lr rc=$?"
python tools/measure_finalize.py > $o/finalize.txt 2> $o/fin.err; echo "fin rc=$?"
python tools/measure_host_path.py > $o/host_path.json 2> $o/host.err; echo "host rc=$?"
bash tools/rehearse_n2.sh > $o/rehearsal_n2.json 2> $o/rehearsal_n2.err; echo "n2 rc=$?"
KMC_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 5 --warmup 3 --total-fasta-bytes 4e9 > $o/rehearsal_n2_strong.json 2> $o/rehearsal_n2_strong.err; echo "n2 strong rc=$?"
for f in bench_n1 bench_n1_k21 bench_n1_k63 bench_n1_pool0_1GB bench_n1_stream rehearsal_n2 rehearsal_n2_strong; do python3 - <<P
import json
try:
    d=json.loads([l for l in open("$o/$f.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("$f", d["value"], d["ms_per_step"], r.get("kernel_ms"), r["frac"], d["scaling"], d["config"].get("exact_full_size_check"), r.get("sort_pipeline",{}).get("frac"))
except Exception as e: print("$f unreadable", e)
P
done
cat $o/lr_mode.txt | tail -1 | cut -c1-600; cat $o/finalize.txt
