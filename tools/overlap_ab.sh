#!/bin/bash
# Launch groups in flight (RWR_WF_OVERLAP = ray queues, RWR_WF_STAGGER), same box, same build: tools/overlap_ab.sh
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --cpu-seconds 0 --config $1 --steps $2 --warmup 2 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1 queues=$RWR_WF_OVERLAP stagger=$RWR_WF_STAGGER', d['ms_per_step'], 'ms')"; }
for i in 1 2; do
  for cfg in cfg3 cfg5; do
    RWR_WF_OVERLAP=1 RWR_WF_STAGGER=0 run $cfg 10
    RWR_WF_OVERLAP=2 RWR_WF_STAGGER=0 run $cfg 10
    RWR_WF_OVERLAP=2 RWR_WF_STAGGER=1 run $cfg 10
    RWR_WF_OVERLAP=3 RWR_WF_STAGGER=1 run $cfg 10
    RWR_WF_OVERLAP=4 RWR_WF_STAGGER=1 run $cfg 10
    RWR_WF_OVERLAP=4 RWR_WF_STAGGER=0 run $cfg 10
  done
done
