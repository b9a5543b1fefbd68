#!/bin/bash
# Memory-pipe counters for the bench kernel (separate --pmc passes, --kernel-trace only).
# usage: tools/pmc_mem.sh <outdir> [bench args...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; echo "pass $name"; timeout -k 5 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --cpu-seconds 0 --steps 100 --warmup 10 $BENCH_ARGS > "$OUT/$name.log" 2>&1 || echo "pass $name failed (see $OUT/$name.log)"; }
BENCH_ARGS="$*"
run ta1  TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
run ta2  TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run sqc  SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_MISSES
run lvl  SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_BUSY_CYCLES
python3 tools/pmc_summary.py "$OUT"
