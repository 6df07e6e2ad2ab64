#!/bin/bash
# The whole GPU side of a round: parity suite, then tools/profile_round.sh (default bench line, kernel statistics, HBM counters, BA split + timeline, driver command).
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$R
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$R.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_$R.log; tail -4 gpurun_out/pytest_$R.log
bash tools/profile_round.sh $R
cp gpurun_out/pytest_$R.log gpurun_out/prof_$R/pytest.log
