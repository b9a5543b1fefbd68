#!/bin/bash
# rocprofv3 kernel stats of one bench config: tools/kstats.sh <tag> <bench args...>   (writes gpurun_out/kstats_<tag>.csv)
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kstats_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --cpu-seconds 0 "$@" > $OUT.log 2>&1
cp $OUT/*/*kernel_stats.csv gpurun_out/kstats_$TAG.csv
cut -d, -f1-8 gpurun_out/kstats_$TAG.csv | sed 's/(.*)//' | cut -c1-160
