#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/final_lite; mkdir -p $o
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $o/tests.log 2>&1; tail -2 $o/tests.log
python bench.py --pool 0 --fasta-bytes 1e9 --steps 5 --warmup 2 > $o/bench_n1_pool0_1GB.json 2> $o/pool0.err; echo "pool0 rc=$?"
python tools/pool_sweep.py --fasta-bytes 1e9 > $o/pool_sweep.jsonl 2> $o/sweep.err; echo "sweep rc=$?"
python tools/measure_lr.py > $o/lr_mode.txt 2> $o/lr.err; echo "lr rc=$?"
python tools/measure_file_highcard.py 2>/dev/null | tail -1 > $o/file_highcard.json; echo "file rc=$?"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_sort -- python3 bench.py --pool 0 --fasta-bytes 1e9 --algo sort --steps 3 --warmup 2 --no-cpu-baseline > $o/bench_sort_prof.json 2> $o/prof.err; echo "prof rc=$?"
python3 - <<P
import json
d=json.loads([l for l in open("$o/bench_n1_pool0_1GB.json") if l.startswith("{")][-1]); print("pool0", d["value"], d["ms_per_step"], d["roofline"]["sort_pipeline"]["frac"])
for l in open("$o/pool_sweep.jsonl"):
    x=json.loads(l); print(x["k"], x["pool"], x["algo_last"], x["step_ms"], "first", x["first_step_ms"])
print(open("$o/lr_mode.txt").read().strip().splitlines()[-1][:400])
print(open("$o/file_highcard.json").read()[:600])
P
