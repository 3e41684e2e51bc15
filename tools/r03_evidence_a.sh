#!/bin/bash
# Round 3, un-profiled evidence (one gpurun call): bench lines, sweep, LR, finalize, host path, rehearsals of the N>1 default.
# Output under gpurun_out/ev3/; the lines to keep are copied to profiles/r03_* by hand afterwards (tools/r03_collect.sh).
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/ev3; mkdir -p $o
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_n1.json 2> $o/bench_n1.err; echo "n1 rc=$?"
python bench.py --k 21 --fasta-bytes 1e9 --seed 1 > $o/bench_n1_k21.json 2> $o/bench_k21.err; echo "k21 rc=$?"
python bench.py --k 63 > $o/bench_n1_k63.json 2> $o/bench_k63.err; echo "k63 rc=$?"
python bench.py --pool 0 --fasta-bytes 1e9 --steps 5 --warmup 2 > $o/bench_n1_pool0_1GB.json 2> $o/bench_pool0.err; echo "pool0 rc=$?"
python bench.py --pool 0 --k 63 --fasta-bytes 1e9 --steps 5 --warmup 2 --no-cpu-baseline > $o/bench_n1_pool0_1GB_k63.json 2> $o/bench_pool0_k63.err; echo "pool0 k63 rc=$?"
python bench.py --algo stream --steps 5 --warmup 2 --no-cpu-baseline > $o/bench_n1_stream.json 2> $o/bench_stream.err; echo "stream rc=$?"
python tools/pool_sweep.py --fasta-bytes 1e9 --pools 10,16,20,26,32,50,100,300,1000,0 > $o/pool_sweep.jsonl 2> $o/pool_sweep.err; echo "sweep rc=$?"
python tools/measure_lr.py > $o/lr_mode.json 2> $o/lr.err; echo "lr rc=$?"
python tools/measure_finalize.py > $o/finalize.txt 2> $o/fin.err; echo "fin rc=$?"
python tools/measure_host_path.py > $o/host_path.json 2> $o/host.err; echo "host rc=$?"
python tools/measure_file_highcard.py > $o/file_highcard.json 2> $o/file_highcard.err; echo "highcard rc=$?"
for i in 1 2 3; do python tools/host_step_times.py 2>/dev/null | tail -1; done > $o/host_step_times.txt; echo "step times rc=$?"
for n in 2 3; do
  KMC_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2963$n \
    bench.py --gpus $n --steps 10 --warmup 5 > $o/rehearsal_n$n.json 2> $o/rehearsal_n$n.err; echo "rehearsal n=$n rc=$?"
done
for f in bench_n1 bench_n1_k21 bench_n1_k63 bench_n1_pool0_1GB bench_n1_pool0_1GB_k63 bench_n1_stream rehearsal_n2 rehearsal_n3; do python3 - <<P
import json
try:
    d=json.loads([l for l in open("$o/$f.json") if l.startswith("{")][-1]); r=d["roofline"]
    print("$f", d["value"], d["ms_per_step"], r.get("kernel_ms"), r["frac"], r.get("frac_of_measured"), d["scaling"], (d["config"].get("exact_full_size_check") or {}).get("bit_exact"), r.get("sort_pipeline",{}).get("frac"), (d.get("weak_scaling") or {}).get("ms_per_step"))
except Exception as e: print("$f unreadable", e)
P
done
python3 -c "
import json
for l in open('$o/pool_sweep.jsonl'):
    if l.startswith('{'):
        d=json.loads(l); print(d['k'], d['pool'], d['algo_last'], d['distinct'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'])
"
tail -1 $o/lr_mode.json | cut -c1-700; cat $o/finalize.txt; cat $o/host_step_times.txt
