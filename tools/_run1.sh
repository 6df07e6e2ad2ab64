cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_sequence_gpu.py -q -m gpu > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log
grep -E "^E |passed|failed|rc=|Error" gpurun_out/t1.log | cut -c1-300 | head -20
DMVIO_HIP_BA_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --batch 64 2>&1 | grep -E "dmvio_hip_ba|GN-iters" | sed 's/.*"ba": /ba: /' | cut -c1-300
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --batch 64 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('BA its/s', d['ba']['value'], d['ba']['ms_per_iter'])"
