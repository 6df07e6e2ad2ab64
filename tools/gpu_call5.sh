#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r04; mkdir -p $O; rm -f $O/kernel_stats_pipeline.md $O/kt2.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu --no-traffic > $O/kt.log 2>&1
python tools/rocprof_summary.py $(find $O/kt -name '*.db' | head -1) > $O/kernel_stats.md 2>> $O/kt.log
rm -rf $O/kt
python bench.py --steps 20 --warmup 3 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout 300 python -m pytest tests/test_bench_contract_gpu.py -x -q 2>&1 | tail -2
head -5 $O/kernel_stats.md | cut -c1-60,130-230
