cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python bench.py --no-cpu --no-ba --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('kernel %.4f ms  frac %.3f pyr %.4f (%.0f GB/s) step %.3f ms value %.0f' % (r['kernel_ms'], r['frac'], r['pyramid_kernel_ms'], r['pyramid_GBps'], d['ms_per_step'], d['value']))"; done
timeout 600 python -m pytest tests/test_tracker_gpu.py -q -m gpu -k pyramid 2>&1 | tail -1
