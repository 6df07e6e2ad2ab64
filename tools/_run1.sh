cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/p_stats gpurun_out/p_fetch gpurun_out/p_write
timeout 600 python bench.py > gpurun_out/bench_full.log 2>&1; tail -1 gpurun_out/bench_full.log | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/p_stats -o stats -- python bench.py --no-cpu > gpurun_out/p_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/p_fetch -o fetch -- python bench.py --no-cpu --steps 3 --warmup 1 --ba-iters 20 > gpurun_out/p_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/p_write -o write -- python bench.py --no-cpu --steps 3 --warmup 1 --ba-iters 20 > gpurun_out/p_write.log 2>&1
DMVIO_HIP_BA_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --batch 64 2>&1 | grep -E "dmvio_hip_ba" > gpurun_out/ba_timing.log
python tools/rocprof_summary.py gpurun_out/p_stats/*.db | head -6 | cut -c1-160
