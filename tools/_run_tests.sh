cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log
grep -E "^E |passed|failed|rc=|Error" gpurun_out/t1.log | cut -c1-300 | head -30
