cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps(d['trace'])); print(d['value'], d['ba']['value'])"
