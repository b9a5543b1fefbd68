#!/bin/bash
# Round-3 evidence: bench lines, kernel stats and PMC passes for every configuration DESIGN.md quotes.
# Writes gpurun_out/profile_r03/; tools/collect_profiles_r03.py copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# usage: tools/profile_round3.sh counters   (kernel stats + PMC passes; then run tools/collect_profiles_r03.py here)
#        tools/profile_round3.sh bench      (the bench lines, which quote the counters just collected)
O=gpurun_out/profile_r03; mkdir -p $O
if [ "$1" = "bench" ]; then
b() { name=$1; shift; python3 bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; tail -c 400 $O/bench_$name.json | head -c 200; echo; }
b default
b driver_cmd --gpus 1 --steps 20 --warmup 5
b cfg2b --cpu-seconds 0 --config cfg2b
b cfg3 --cpu-seconds 0 --config cfg3
b cfg4 --cpu-seconds 0 --config cfg4
b cfg5_1gpu --cpu-seconds 0 --config cfg5
b cfg2_force_dist --cpu-seconds 0 --force-dist
b cfg2_force_dist_fif1 --cpu-seconds 0 --force-dist --frames-in-flight 1
b cfg2_force_dist_fif2 --cpu-seconds 0 --force-dist --frames-in-flight 2
b cfg5_force_dist --cpu-seconds 0 --force-dist --config cfg5
b loop --cpu-seconds 0 --config loop
b loop_fif1 --cpu-seconds 0 --config loop --frames-in-flight 1
ls $O | head -50
exit 0
fi
k() { name=$1; shift; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 bench.py --cpu-seconds 0 "$@" > $O/trace_$name.log 2>&1; cp $O/trace_$name/*/*kernel_stats.csv $O/kernel_stats_$name.csv; }
k cfg2 --steps 500 --warmup 100
k cfg2_one_in_flight --steps 500 --warmup 100 --frames-in-flight 1
k cfg4_one_in_flight --config cfg4 --steps 20 --warmup 4 --frames-in-flight 1
k cfg3 --config cfg3 --steps 4 --warmup 2 --skip-serial
k cfg4 --config cfg4 --steps 10 --warmup 2 --skip-serial
k cfg5 --config cfg5 --steps 5 --warmup 2 --skip-serial
tools/pmc2.sh $O/pmc_cfg2 "--steps 200 --warmup 20" inst cyc fetch write > $O/pmc_cfg2.txt 2>&1
tools/pmc2.sh $O/pmc_cfg2b "--config cfg2b --steps 200 --warmup 20" inst cyc fetch write > $O/pmc_cfg2b.txt 2>&1
# (thr: SQ_THREAD_CYCLES_VALU beside SQ_ACTIVE_INST_VALU — how many lanes the vector instructions had live)
tools/pmc2.sh $O/pmc_cfg3 "--config cfg3 --steps 2 --warmup 2 --skip-serial" inst cyc fetch write thr > $O/pmc_cfg3.txt 2>&1
tools/pmc2.sh $O/pmc_cfg4 "--config cfg4 --steps 3 --warmup 2 --skip-serial" inst cyc fetch write thr > $O/pmc_cfg4.txt 2>&1
tools/pmc2.sh $O/pmc_cfg5 "--config cfg5 --steps 2 --warmup 2 --skip-serial" inst cyc fetch write thr > $O/pmc_cfg5.txt 2>&1
ls $O | head -50
