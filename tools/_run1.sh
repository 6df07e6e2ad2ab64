cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log
grep -E "^E |passed|failed|rc=|Error" gpurun_out/t1.log | cut -c1-300 | head -20
for b in 1 4 8 16 31 64 128; do MB_BATCH=$b timeout 120 python tools/microbench.py 2>&1 | grep track_lm | tail -1; done
timeout 300 python bench.py --no-cpu --no-ba --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('B=1024 kernel %.4f ms  frac %.3f  step %.3f ms value %.0f' % (r['kernel_ms'], r['frac'], d['ms_per_step'], d['value']))"
