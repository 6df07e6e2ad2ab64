#!/bin/bash
# rocprofv3 kernel summary of the batched BA probe (run through gpurun from the repo root): tools/ba_batch_prof.sh <Ws> <out.md>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profb
timeout 120 rocprofv3 --kernel-trace -d /tmp/profb -o p -- python $GRAFT_REPO_ROOT/tools/ba_batch_probe.py ${1:-16} > /tmp/profb.log 2>&1 < /dev/null
DB=$(find /tmp/profb -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $DB > $GRAFT_REPO_ROOT/${2:-gpurun_out/ba_batch_prof.md} < /dev/null
grep -E "^W=|k_ba_solve timeline" /tmp/profb.log
