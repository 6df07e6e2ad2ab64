#!/bin/bash
# Hardware-counter passes for the batched bundle adjustment at W windows per launch (tools/ba_batch_probe.py): what the CUs, the vector L1 (TCP) and the L2 (TCC) do during
# k_ba_linearize_b1 / k_ba_accumulate_b / k_ba_stitch_b.  One rocprofv3 run per counter group (--kernel-trace only, no other tracing); summaries land in
# gpurun_out/prof_<round>/counters_ba_<group>.md.   usage: tools/profile_counters_ba.sh r06 [W]
R=${1:-r06}; W=${2:-64}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$R; mkdir -p $O
run() {  # name counters...
  local name=$1; shift
  if [ -n "$ONLY" ] && [[ " $ONLY " != *" $name "* ]]; then return; fi
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/cb_$name -o c -- python tools/ba_batch_probe.py $W > $O/cb_$name.log 2>&1
  python tools/rocprof_summary.py $(find $O/cb_$name -name '*.db' | head -1) --counters 2>> $O/cb_$name.log | grep -E "kernel|---|k_ba_linearize_b1|k_ba_accumulate_b|k_ba_stitch_b|k_ba_solve" > $O/counters_ba_$name.md
  rm -rf $O/cb_$name
  cat $O/counters_ba_$name.md | cut -c1-60,100-220 | head -40
}
run sq_insts SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES
run sq_time SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq_occ SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
