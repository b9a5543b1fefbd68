#!/bin/bash
# per-lane trace kernel: work items per launch (RWR_WF_LANE_ITEMS) with waves drawing 64-ray pieces of an item on their own
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --cpu-seconds 0 --config $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$2', '$1', d['ms_per_step'], 'ms', d.get('ms_per_frame_one_in_flight'))"; }
timeout -k 10 300 python3 -m pytest tests/test_gpu_path.py -x -q -m gpu 2>&1 | tail -1
for cfg in cfg4 cfg5; do
  RWR_HIP_LIB=$GRAFT_REPO_ROOT/rust-wgpu-raytracing_amd/lib_b/librwr_hip.so run $cfg "B(before)"
  for n in 16384 8192 4096 2048; do RWR_WF_LANE_ITEMS=$n run $cfg "items=$n"; done
  RWR_HIP_LIB=$GRAFT_REPO_ROOT/rust-wgpu-raytracing_amd/lib_b/librwr_hip.so run $cfg "B(before)"
done
