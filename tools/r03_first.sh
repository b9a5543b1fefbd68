#!/bin/bash
# round 3, first GPU pass: new dist tests, whole GPU suite, bench lines (default, driver command, gather rehearsal, loop)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03_first; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu > $O/pytest_dist.log 2>&1; echo "dist rc=$?"; tail -5 $O/pytest_dist.log
b() { name=$1; shift; timeout -k 10 300 python3 bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name rc=$?"; python3 - $O/bench_$name.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print({k:d.get(k) for k in ("value","ms_per_step","ms_per_frame_one_in_flight","ms_per_frame_render_only")}, d["config"].get("gathered_frame_ok"), r.get("frac"), r.get("counters_stale"), d.get("loop"))
except Exception as e: print("no line", e)
P
}
b default --cpu-seconds 0
b driver_cmd --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0
b cfg2_force_dist --cpu-seconds 0 --force-dist
b cfg2_force_dist_fif1 --cpu-seconds 0 --force-dist --frames-in-flight 1
b cfg2_force_dist_fif3 --cpu-seconds 0 --force-dist --frames-in-flight 3
b cfg5_force_dist --cpu-seconds 0 --force-dist --config cfg5
b loop --cpu-seconds 0 --config loop
b loop_fif1 --cpu-seconds 0 --config loop --frames-in-flight 1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu rc=$?"; tail -5 $O/pytest_gpu.log
