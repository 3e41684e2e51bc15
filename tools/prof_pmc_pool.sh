#!/bin/bash
# PMC counters (separate passes, no other trace domain) of one point of the cardinality sweep, summed per kernel.
# usage: bash tools/prof_pmc_pool.sh <tag> <pool> <k> [algo]
tag=$1; pool=$2; k=$3; algo=${4:-auto}
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/pmc_${tag}
mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAVES" \
  "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -- python3 tools/pool_sweep.py --pools $pool --ks $k --steps 2 --algo $algo > $out/p$i.log 2>&1
  echo "pmc$i rc=$?"
done
python3 - <<P
import csv,glob,collections,json
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$out/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0][:44]
        agg[n][r["Counter_Name"]]+=float(r["Counter_Value"]); calls[n][r["Counter_Name"]]+=1
res={}
for n,c in agg.items():
    res[n]={k:(v/max(1,calls[n][k])) for k,v in c.items()}; res[n]["calls"]=max(calls[n].values())
json.dump(res,open("$out/summary.json","w"),indent=1)
keys=["SQ_BUSY_CYCLES","SQ_WAVES","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_WAIT_INST_ANY","SQ_WAVE_CYCLES","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","FETCH_SIZE","WRITE_SIZE"]
for n,c in sorted(res.items(), key=lambda kv:-kv[1].get("SQ_BUSY_CYCLES",0))[:8]:
    print(n, "calls", c["calls"])
    print("   "+"  ".join(f"{k.replace('SQ_','')}={c.get(k,0):.3g}" for k in keys))
P
