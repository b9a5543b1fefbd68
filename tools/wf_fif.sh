#!/bin/bash
# Wavefront frames with one and two frame slots (a slot owns a set of accumulators and ray queues), same box: tools/wf_fif.sh
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --cpu-seconds 0 "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$*', d['ms_per_step'], 'ms', d.get('ms_per_frame_one_in_flight'))"; }
for i in 1 2; do
  for cfg in cfg4 cfg5 cfg3; do
    run --config $cfg --frames-in-flight 1
    run --config $cfg --frames-in-flight 2
  done
done
