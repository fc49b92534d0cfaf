#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c5; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=12 > $O/gputests.log 2>&1; tail -22 $O/gputests.log
T2H_FORCE_DIST=1 python bench.py --steps 5 --warmup 2 > $O/bench_dist1.json 2> $O/bench_dist1.err; tail -c 600 $O/bench_dist1.json; tail -2 $O/bench_dist1.err
