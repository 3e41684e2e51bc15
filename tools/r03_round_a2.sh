#!/bin/bash
# the part of tools/r03_round_a.sh behind the test suite, plus the tests named on the command line
tag=${1:-r03a}; filt=${2:-}
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
if [ -n "$filt" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$filt" > gpurun_out/${tag}_tests.log 2>&1
  rc=$?; echo "tests rc=$rc"; tail -8 gpurun_out/${tag}_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err
echo "bench rc=$?"; tail -3 gpurun_out/${tag}_bench_n1.err; cat gpurun_out/${tag}_bench_n1.json | cut -c1-1800
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-cold --no-exact-check --no-read-peak > /dev/null 2> gpurun_out/${tag}_trace.err
python3 - <<P
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/${tag}_trace/*/*_kernel_trace.csv"):
    rows+=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:48]) for r in csv.DictReader(open(f))]
for f in glob.glob("gpurun_out/${tag}_trace/*/*_memory_copy_trace.csv"):
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
    bench.py --gpus $n --steps 10 --warmup 5 > gpurun_out/${tag}_rehearsal_n${n}.json 2> gpurun_out/${tag}_rehearsal_n${n}.err
  echo "rehearsal n=$n rc=$?"; tail -3 gpurun_out/${tag}_rehearsal_n${n}.err; cut -c1-1200 gpurun_out/${tag}_rehearsal_n${n}.json
done
