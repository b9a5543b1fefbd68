#!/bin/bash
# A/B kernel trace of two builds: tools/ab_trace.sh <config> <old lib path>
set -e
CFG=${1:-cfg4}; OLD=${2:-gpurun_in/librwr_hip_old.so}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ab_$CFG; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/new" -- python3 bench.py --cpu-seconds 0 --config $CFG --steps 3 --warmup 1 > "$OUT/new.log" 2>&1
export RWR_HIP_LIB=$OLD
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/old" -- python3 bench.py --cpu-seconds 0 --config $CFG --steps 3 --warmup 1 > "$OUT/old.log" 2>&1
echo NEW; cut -d, -f1-4 "$OUT"/new/*/*kernel_stats.csv | head -8
echo OLD; cut -d, -f1-4 "$OUT"/old/*/*kernel_stats.csv | head -8
