#!/bin/bash
# identity + RAS counters of the box, then the walk stress; appends to gpurun_out/stress_boxes.log
out=${GRAFT_REPO_ROOT:-.}/gpurun_out/stress_boxes.log
mkdir -p $(dirname $out)
{
echo "==== $(date -u +%FT%TZ) host $(hostname)"
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id"
rocm-smi --showrasinfo all 2>/dev/null | grep -v -E "^=|^$" | head -40
for s in 15 51 52 53 54 55; do timeout -k 10 300 python tools/stress_walk.py fresh $s 500 2>&1 | grep -E "MISMATCH|fails"; done
rocm-smi --showrasinfo all 2>/dev/null | grep -i -E "UE|CE|uncorrect|correct" | head -20
} 2>&1 | tee -a $out | tail -40
