#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3; do
  python tools/host_step_times.py 2>/dev/null | tail -1
  KMC_WALK_EVENT_RECORDS=1 python tools/host_step_times.py 2>/dev/null | tail -1 | sed 's/^/EVREC /'
done
python tools/host_step_times.py 1e9 2>/dev/null | tail -1
