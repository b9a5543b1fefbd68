#!/bin/bash
# FETCH_SIZE calibration for 32-byte gathers (tools/ubench/gather32.hip): time, then the counter.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gather_calib; rm -rf $O; mkdir -p $O
tools/ubench/gather32 > $O/time.txt 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc -- tools/ubench/gather32 > $O/pmc.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("gpurun_out/gather_calib/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in rows.items():
    print(k, "FETCH_SIZE (KB) per launch:", [round(x) for x in v])
PY
cat $O/time.txt
