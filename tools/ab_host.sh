#!/bin/bash
# same-box A/B of the host path (file -> table) between the round-1 tree and the current one
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do for t in ab_r01 .; do (cd $t && python tools/measure_host_path.py 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t', d['host_buffers']['GBps'], d['fasta_file_1GB'])"); done; done
