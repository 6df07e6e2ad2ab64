#!/bin/bash
# Short GPU check of a round in progress: the GPU parity suite, the default bench line and the kernel statistics of the same command.
#   usage (through gpurun, from the repo root): tools/quick_round.sh r04a
R=${1:-r04a}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/quick_$R; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"
timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu --no-traffic > $O/kt.log 2>&1
python tools/rocprof_summary.py $(find $O/kt -name '*.db' | head -1) > $O/kernel_stats.md 2>> $O/kt.log
rm -rf $O/kt
head -30 $O/kernel_stats.md | cut -c1-200
tail -c 1500 $O/bench_n1.json
