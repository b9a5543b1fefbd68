#!/bin/bash
# PMC passes (one counter group per rocprofv3 run, --kernel-trace only).  usage: tools/pmc2.sh <outdir> "<bench args>" group1 group2 ...
# groups: inst cyc fetch write valu lds
set -e
OUT=$1; BENCH_ARGS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
declare -A G
G[inst]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH"
G[cyc]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
G[fetch]="FETCH_SIZE GRBM_GUI_ACTIVE"
G[write]="WRITE_SIZE"
G[valu]="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS"
G[lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS"
G[thr]="SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
for g in "$@"; do
  echo "pass $g"
  timeout -k 5 280 rocprofv3 --kernel-trace --pmc ${G[$g]} --output-format csv -d "$OUT/$g" -- python3 bench.py --cpu-seconds 0 $BENCH_ARGS > "$OUT/$g.log" 2>&1 || { echo "pass $g failed"; tail -5 "$OUT/$g.log"; }
done
python3 tools/pmc_summary.py "$OUT"
