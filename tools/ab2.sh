#!/bin/bash
# A/B of two builds on one box over several configurations: tools/ab2.sh cfg4 cfg5 ...   (lib/ = A, lib_b/ = B; two rounds each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --cpu-seconds 0 $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$2', '$1', d['ms_per_step'], 'ms', d.get('ms_per_frame_one_in_flight'))"; }
for cfg in "$@"; do for i in 1 2; do
  run "--config $cfg" A
  RWR_HIP_LIB=$GRAFT_REPO_ROOT/rust-wgpu-raytracing_amd/lib_b/librwr_hip.so run "--config $cfg" B
done; done
