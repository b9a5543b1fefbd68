#!/bin/bash
# wavefront check: parity tests, then cfg3 / cfg4 timings (optionally per-kernel stats)
set -e
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_gpu_path.py tests/test_multi_material.py -x -q 2>&1 | tail -15
for cfg in cfg3 cfg4; do
  python3 bench.py --cpu-seconds 0 --config $cfg | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$cfg', d['ms_per_step'], 'ms', d['value'], 'Mray/s')"
done
RWR_WF_PACKET_FILL=2 python3 bench.py --cpu-seconds 0 --config cfg4 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg4 no packets', d['ms_per_step'], 'ms')"
RWR_WF_PACKET_FILL=0.6 python3 bench.py --cpu-seconds 0 --config cfg4 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg4 fill 0.6', d['ms_per_step'], 'ms')"
RWR_WF_GROUP=32 python3 bench.py --cpu-seconds 0 --config cfg3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg3 group 32', d['ms_per_step'], 'ms')"
RWR_WF_GROUP=8 python3 bench.py --cpu-seconds 0 --config cfg3 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cfg3 group 8', d['ms_per_step'], 'ms')"
tools/kstats.sh cfg3 --config cfg3 --steps 3 --warmup 1 > /dev/null
tools/kstats.sh cfg4 --config cfg4 --steps 3 --warmup 1 > /dev/null
python3 - <<'PY'
import csv
for f in ('gpurun_out/kstats_cfg3.csv','gpurun_out/kstats_cfg4.csv'):
    print(f)
    for r in csv.DictReader(open(f)):
        if 'wf' in r['Name'] or 'bin' in r['Name']:
            print("  %-44s calls %5s avg %12.1f us"%(r['Name'].split('(')[0][:44], r['Calls'], float(r['AverageNs'])/1e3))
PY
