#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/call3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_io_gpu.py tests/test_tracker_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 300 python tools/time_pyramids.py > $O/time_pyramids.log 2>&1; cat $O/time_pyramids.log | grep -v Warn
timeout 300 python bench.py --no-cpu --no-traffic --no-sweep --no-pcie --no-ba --steps 20 --warmup 3 > $O/bench_short.json 2> $O/bench_short.err; tail -c 1200 $O/bench_short.json
