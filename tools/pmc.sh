#!/bin/bash
# PMC passes for the bench kernel (each counter group in its own rocprofv3 run,
# --kernel-trace only; never combined with other trace domains).
# usage: tools/pmc.sh <outdir> [bench args...]
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; echo "pass $name"; timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --cpu-seconds 0 ${PMC_STEPS:---steps 200 --warmup 20} $BENCH_ARGS > "$OUT/$name.log" 2>&1; }
BENCH_ARGS="$*"
run inst  SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH
run cyc   SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
run valu  SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS
python3 tools/pmc_summary.py "$OUT"
