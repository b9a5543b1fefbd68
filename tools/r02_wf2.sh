#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_gpu_path.py tests/test_multi_material.py -x -q 2>&1 | tail -8
for c in cfg3 cfg4; do RWR_WF_STATS=1 python3 bench.py --cpu-seconds 0 --config $c 2>&1 | grep -v amdgpu | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$c', d['ms_per_step'], 'ms', d['value'], 'Mray/s')
    elif 'wavefront' in l: print(l.strip())"; done
for c in cfg3 cfg4; do tools/kstats.sh $c --config $c --steps 3 --warmup 1 > /dev/null; python3 - <<PY
import csv
print('$c')
for r in csv.DictReader(open('gpurun_out/kstats_$c.csv')):
    if 'wf' in r['Name'] or 'bin' in r['Name']:
        print("  %-44s calls %5s avg %12.1f us"%(r['Name'].split('(')[0][:44], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
