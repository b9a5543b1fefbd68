#!/bin/bash
# one-knob-at-a-time sweep around the defaults: tools/knob_sweep.sh cfg5 "RWR_WF_ZSPLIT=4" "RWR_BVH_LEAF=3" "--frames-in-flight 3" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cfg=$1; shift
run() { env $1 python3 bench.py --cpu-seconds 0 --config $cfg $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg', '$1 $2', d['ms_per_step'], 'ms', d.get('ms_per_frame_one_in_flight'))"; }
run "A=0" ""
for k in "$@"; do
  case "$k" in --*) run "A=0" "$k";; *) run "$k" "";; esac
done
run "A=0" ""
