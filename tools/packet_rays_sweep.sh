#!/bin/bash
# which pools are traced as packets: RWR_WF_PACKET_RAYS (pools of at least that many rays, whatever their origins' extent)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python3 bench.py --cpu-seconds 0 --config $1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$2', '$1', d['ms_per_step'], 'ms', d.get('ms_per_frame_one_in_flight'))"; }
for cfg in "$@"; do
  run $cfg "rays=off"
  for n in 2000 4000 6000 8000 12000 16000 24000; do RWR_WF_PACKET_RAYS=$n run $cfg "rays=$n"; done
  run $cfg "rays=off"
done
