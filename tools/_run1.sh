cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_tracker_gpu.py -q -m gpu 2>&1 | tail -2
for b in 1 31; do MB_BATCH=$b timeout 120 python tools/microbench.py 2>&1 | grep track_lm | tail -1; done
timeout 300 python bench.py --no-cpu --no-ba --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('B=1024 kernel %.4f ms  frac %.3f  step %.3f ms value %.0f' % (r['kernel_ms'], r['frac'], d['ms_per_step'], d['value']))"
