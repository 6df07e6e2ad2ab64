#!/bin/bash
# kernel timeline of one dmvio_hip_ba_optimize_batch call at W windows (run through gpurun from the repo root): tools/ba_batch_timeline.sh <W> <out.txt>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/proft
timeout 120 rocprofv3 --kernel-trace -d /tmp/proft -o p -- python $GRAFT_REPO_ROOT/tools/ba_batch_probe.py ${1:-16} > /tmp/proft.log 2>&1 < /dev/null
DB=$(find /tmp/proft -name "*.db" | head -1)
[ -n "$DB" ] && timeout 60 python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py $DB k_ba_solve 70 30 > $GRAFT_REPO_ROOT/${2:-gpurun_out/ba_batch_timeline.txt} < /dev/null
grep -E "^W=" /tmp/proft.log
