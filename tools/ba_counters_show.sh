#!/bin/bash
# prints the counter summaries of tools/profile_counters_ba.sh for the batched BA kernels: kernel, counter, dispatches, avg, max
cd ${GRAFT_REPO_ROOT:-.}/gpurun_out/prof_${1:-r06}
for f in counters_ba_*.md; do echo "== $f"; grep -E "k_ba_linearize_b1|k_ba_accumulate_b|k_ba_stitch_b" $f | awk -F'|' '{split($2,a,"("); printf "%-24s %-42s n=%s avg=%s max=%s\n", a[1], $3, $4, $5, $7}'; done
