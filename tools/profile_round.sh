#!/bin/bash
# Regenerates the summaries committed under profiles/ (run on the GPU box through gpurun from the repo root).
#   usage: tools/profile_round.sh r01
# Counter passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa tracing).
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$R; rm -rf $O; mkdir -p $O
python tools/publish_round.py --hash > $O/csrc_hash.txt   # what is being profiled (tools/publish_round.py refuses summaries of other sources)
t0=$(date +%s)
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
t1=$(date +%s); echo "default bench.py wall seconds: $((t1 - t0))" > $O/bench_wall.txt
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu --no-traffic > $O/kt.log 2>&1
python tools/rocprof_summary.py $(find $O/kt -name '*.db' | head -1) > $O/kernel_stats.md 2>> $O/kt.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o pmc -- python bench.py --no-cpu --no-traffic --no-sweep --no-pcie --steps 3 --warmup 1 --ba-iters 20 > $O/pmc_$C.log 2>&1
  python tools/rocprof_summary.py $(find $O/pmc_$C -name '*.db' | head -1) --counters > $O/pmc_$C.md 2>> $O/pmc_$C.log
done
DMVIO_HIP_BA_TIMING=1 python bench.py --no-cpu --no-traffic --no-sweep --no-pcie --steps 3 --warmup 1 > $O/ba_timing.log 2>&1
rocprofv3 --kernel-trace -d $O/ba_kt -o kt -- python tools/ba_loop.py 300 > $O/ba_loop.log 2>&1
python tools/rocprof_timeline.py $(find $O/ba_kt -name '*.db' | head -1) k_ba_point_sums 34 2 > $O/ba_timeline.txt 2>> $O/ba_loop.log
rm -rf $O/ba_kt
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python bench.py --steps 20 --warmup 3 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
bash tools/ba_batch_prof.sh 16 $O/ba_batch_kernels_w16.md > $O/ba_batch_probe_w16.log 2>&1 < /dev/null
bash tools/ba_batch_prof.sh 64 $O/ba_batch_kernels_w64.md > $O/ba_batch_probe_w64.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
tail -c 600 $O/bench_n1.json; cat $O/bench_wall.txt
