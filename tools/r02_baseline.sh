#!/bin/bash
# round-2 baseline on the GPU box: tests, the driver's bench command, the default bench, cfg3
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_base; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
tail -1 $O/bench_driver.json
python3 bench.py --cpu-seconds 0 > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json
python3 bench.py --cpu-seconds 0 --config cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
tail -1 $O/bench_cfg3.json
python3 bench.py --cpu-seconds 0 --config cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
tail -1 $O/bench_cfg4.json
