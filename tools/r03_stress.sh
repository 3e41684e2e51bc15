#!/bin/bash
# randomised parity stress with seeds the suite does not use (tools/stress_sort_lr.py; ~6 minutes)
cd ${GRAFT_REPO_ROOT:-.}
STRESS_SEED=${1:-777} timeout -k 10 200 python tools/stress_sort_lr.py 150 2>&1 | tail -3
STRESS_SEED=$(( ${1:-777} + 1 )) STRESS_MAX_REC=40000 timeout -k 10 300 python tools/stress_sort_lr.py 200 2>&1 | tail -3
timeout -k 10 200 python tools/stress_walk.py overflow $(( ${1:-777} + 2 )) 2>&1 | tail -3
KMC_STRESS_SEED=$(( ${1:-777} + 3 )) KMC_STRESS_CASES=48 timeout -k 10 400 python -m pytest tests -m gpu -x -q -k planner_randomised 2>&1 | tail -3
