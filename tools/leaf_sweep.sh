#!/bin/bash
# BVH leaf size sweep on one box: tools/leaf_sweep.sh
cd "$GRAFT_REPO_ROOT"
for cfg in cfg3 cfg4; do
  for leaf in 1 2 3 4; do
    RWR_BVH_LEAF=$leaf python3 bench.py --cpu-seconds 0 --config $cfg --steps 10 --warmup 2 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg leaf=$leaf', d['ms_per_step'], 'ms')"
  done
done
