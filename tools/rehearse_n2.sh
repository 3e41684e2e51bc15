#!/bin/bash
# Rehearsal of bench.py --gpus 2 on a ONE-GPU box: two ranks share the card, gloo carries the
# collective (RCCL refuses two ranks on one device).  Checks the N > 1 code path end to end
# (torch-stream ctxs, slab all-gather, owner merge, reduce accounting); its timing says nothing
# about RCCL.  usage: tools/rehearse_n2.sh [bench args]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
KMC_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 2 --steps 5 --warmup 3 --fasta-bytes 2e9 "$@"
