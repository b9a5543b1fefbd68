#!/bin/bash
# A/B of two builds on one box: tools/ab.sh "<bench args>"   (lib/ vs lib_b/)
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --cpu-seconds 0 $1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$2', '$1', d['ms_per_step'], 'ms')"; }
for i in 1 2; do
  run "$1" A
  RWR_HIP_LIB=$GRAFT_REPO_ROOT/rust-wgpu-raytracing_amd/lib_b/librwr_hip.so run "$1" B
done
