cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tracker_gpu.py tests/test_io_gpu.py -q -m gpu > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log
grep -E "^E |passed|failed|rc=|Error" gpurun_out/t1.log | cut -c1-300 | head -20
for b in 1024; do timeout 300 python bench.py --no-cpu --no-ba --batch $b --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('B=%d kernel %.4f ms  frac %.3f  pyramid %.4f ms %.0f GB/s step %.3f ms value %.0f' % (d['config']['frames_per_step_per_gpu'], r['kernel_ms'], r['frac'], r['pyramid_kernel_ms'], r['pyramid_GBps'], d['ms_per_step'], d['value']))"; done
