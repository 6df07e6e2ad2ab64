cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log
grep -E "^E |passed|failed|rc=" gpurun_out/t1.log | cut -c1-300 | head -30
timeout 300 python bench.py --no-cpu --no-ba 2>&1 | tail -1 | cut -c1-1800
MB_BATCH=1 timeout 120 python tools/microbench.py 2>&1 | grep track_lm | tail -1
