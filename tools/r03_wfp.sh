#!/bin/bash
# k_wf_primary without spills: parity tests, then A (lib/) against B (lib_b/: the build before) on the wavefront configurations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_wfp; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_path.py tests/test_gpu_frames_in_flight.py tests/test_gpu_normal_map.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() { python3 bench.py --cpu-seconds 0 $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$2', '$1', d['ms_per_step'], 'ms', d.get('ms_per_frame_one_in_flight'))"; }
for cfg in cfg3 cfg4 cfg5; do for i in 1 2; do
  run "--config $cfg" A
  RWR_HIP_LIB=$GRAFT_REPO_ROOT/rust-wgpu-raytracing_amd/lib_b/librwr_hip.so run "--config $cfg" B
done; done
