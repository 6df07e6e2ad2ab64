#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/call4; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_dropin_gpu.py -x -q -s -k "resident" > $O/pytest_dropin.log 2>&1; echo "pytest dropin rc $?"; grep -E "adapter per keyframe|hand-over split|write-back split|with multiThreading|real marginalizePointsF|passed|failed|Error|assert" $O/pytest_dropin.log | head -20
