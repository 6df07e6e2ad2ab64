#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/call2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_io_gpu.py -x -q > $O/pytest_io.log 2>&1; echo "pytest io rc $?"; tail -4 $O/pytest_io.log
timeout 300 python tools/time_pyramids.py > $O/time_pyramids.log 2>&1; cat $O/time_pyramids.log | grep -v Warn
DMVIO_HIP_BA_TIMING=1 timeout 300 python tools/ba_loop.py 300 > $O/ba_loop.log 2>&1; grep -E "GN-iter|optimize\(|dmvio_hip_ba\]" $O/ba_loop.log | head -20
