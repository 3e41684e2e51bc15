#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "finalize_async or poll_and_forget or rccl or sample_fasta" 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03k_bench_n1.json 2> gpurun_out/r03k_bench_n1.err
echo "bench rc=$?"; tail -2 gpurun_out/r03k_bench_n1.err; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r03k_bench_n1.json') if l.startswith('{')][-1]); r=d['roofline']; print('n1', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_of_measured'))"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r03k_trace -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-cold --no-exact-check --no-read-peak > /dev/null 2> gpurun_out/r03k_trace.err
python3 - <<P
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/r03k_trace/*/*_kernel_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:48]) for r in csv.DictReader(open(f))]
for f in glob.glob("gpurun_out/r03k_trace/*/*_memory_copy_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")) for r in csv.DictReader(open(f))]
rows.sort()
walk=[i for i,r in enumerate(rows) if "kmc_walk_kernel" in r[2]]
if len(walk)>=3:
    i0=walk[-3]; i1=walk[-2]
    t0=rows[i0][0]; prev=t0
    print("== one step (from a walk kernel start to the next)")
    for s,e,n in rows[i0:i1+1]:
        print(f"{(s-t0)/1e3:8.1f} gap {(s-prev)/1e3:6.1f} dur {(e-s)/1e3:7.1f} {n}")
        prev=e
P
for n in 2 3; do
  KMC_BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n \
    bench.py --gpus $n --steps 10 --warmup 5 > gpurun_out/r03k_rehearsal_n${n}.json 2> gpurun_out/r03k_rehearsal_n${n}.err
  echo "rehearsal n=$n rc=$?"; grep -v "Gloo\|socket" gpurun_out/r03k_rehearsal_n${n}.err | tail -3; grep "^{" gpurun_out/r03k_rehearsal_n${n}.json | cut -c1-400
done
python tools/pool_sweep.py --fasta-bytes 1e9 --pools 16,20,50 --ks 63,31 --steps 4 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['k'], d['pool'], d['algo_last'], d['distinct'], 'step_ms', d['step_ms'], 'kern_ms', d['count_kernels_ms'], 'direct', d['direct_share'], 'first', d['first_step_ms'], d['first_step_algo'])
"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03k_prof_p20 -- python3 tools/pool_sweep.py --fasta-bytes 1e9 --pools 20 --ks 63 --steps 4 > /dev/null 2> /dev/null
python3 - <<P
import csv,glob
for f in glob.glob("gpurun_out/r03k_prof_p20/*/*_kernel_trace.csv")[:1]:
    rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:46]) for r in csv.DictReader(open(f))]
    rows.sort()
    walk=[i for i,r in enumerate(rows) if "kmc_walk_kernel" in r[2]]
    i0=walk[-2]; i1=walk[-1]; t0=rows[i0][0]; prev=t0
    print("== pool 20 k=63 step")
    for s,e,n in rows[i0:i1+1]:
        print(f"{(s-t0)/1e3:9.1f} gap {(s-prev)/1e3:7.1f} dur {(e-s)/1e3:8.1f} {n}")
        prev=e
P
