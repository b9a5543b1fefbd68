#!/bin/bash
# Shares of a tile's samples in the primary stage on frames that show little, same box: tools/zsplit_sweep.sh
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --cpu-seconds 0 --config $1 --steps $2 --warmup 2 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1 zsplit=$RWR_WF_ZSPLIT', d['ms_per_step'], 'ms')"; }
for i in 1 2; do
  for cfg in cfg4 cfg5; do
    for z in 0 2 4 8 16 32; do RWR_WF_ZSPLIT=$z run $cfg 10; done
  done
done
