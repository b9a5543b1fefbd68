#!/bin/bash
# One-rank rehearsal of the gather with one and two frames in flight, same box: tools/dist_fif.sh
cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --cpu-seconds 0 --force-dist "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$*', d['ms_per_step'], 'ms', 'render only', d.get('ms_per_frame_render_only'), 'ok', d['config'].get('gathered_frame_ok'))"; }
for i in 1 2; do
  run --frames-in-flight 1
  run --frames-in-flight 2
  run --config cfg5 --frames-in-flight 1
  run --config cfg5 --frames-in-flight 2
done
