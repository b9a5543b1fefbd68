#!/bin/bash
# Produces the rocprofv3 evidence for one round under gpurun_out/profile_<tag>/ (copy the
# summaries you want judged into profiles/).  usage: tools/profile_round.sh r01
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile_$TAG
mkdir -p "$OUT"
# 1. the bench line itself (default command the driver runs)
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -1 "$OUT/bench_default.json"
# 2. kernel trace + stats of the same command (shorter run, CPU baseline off)
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --cpu-seconds 0 --steps 500 > "$OUT/trace.log" 2>&1
cp "$OUT"/trace/*/*kernel_stats.csv "$OUT/kernel_stats_cfg2.csv"
# 2b. the same with one frame in flight: lone launches, the duration bench.py reports as launch_us_serial
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_fif1" -- python3 bench.py --cpu-seconds 0 --steps 500 --frames-in-flight 1 > "$OUT/trace_fif1.log" 2>&1
cp "$OUT"/trace_fif1/*/*kernel_stats.csv "$OUT/kernel_stats_cfg2_one_in_flight.csv"
# 3. PMC passes (counters only, with --kernel-trace; one group per run)
tools/pmc.sh "$OUT/pmc_cfg2" > "$OUT/pmc_cfg2.txt" 2>&1
# 4. wavefront config (cfg3) kernel stats
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_cfg3" -- python3 bench.py --cpu-seconds 0 --config cfg3 --steps 3 --warmup 1 > "$OUT/trace_cfg3.log" 2>&1
cp "$OUT"/trace_cfg3/*/*kernel_stats.csv "$OUT/kernel_stats_cfg3.csv"
cat "$OUT/kernel_stats_cfg2.csv"
cat "$OUT/kernel_stats_cfg3.csv"
