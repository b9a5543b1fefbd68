#!/bin/bash
# Samples per launch group with two groups in flight, same box: tools/group_sweep.sh
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --cpu-seconds 0 --config $1 --steps $2 --warmup 2 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1 group=$RWR_WF_GROUP', d['ms_per_step'], 'ms')"; }
for i in 1 2; do
  for cfg in cfg3 cfg4 cfg5; do
    for g in 8 16 32; do RWR_WF_GROUP=$g run $cfg 10; done
  done
done
