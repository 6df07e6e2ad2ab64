#!/bin/bash
# Hardware-counter passes for the dominant kernel (k_track_lm, batch 1024): what the CUs, the vector L1 (TCP), the TLB and the L2 (TCC)
# do during the launch.  One rocprofv3 run per counter group (--kernel-trace only, no other tracing); summaries land in
# gpurun_out/prof_<round>/counters_<group>.md.   usage: tools/profile_counters.sh r01
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$R; mkdir -p $O
run() {  # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $O/c_$name -o c -- python bench.py --no-cpu --no-ba --no-traffic --no-sweep --no-pcie --steps 3 --warmup 1 > $O/c_$name.log 2>&1
  python tools/rocprof_summary.py $(find $O/c_$name -name '*.db' | head -1) --counters 2>> $O/c_$name.log | grep -E "kernel|---|k_track_lm<256|k_build_pyramids" > $O/counters_$name.md
  rm -rf $O/c_$name
  cat $O/counters_$name.md | cut -c1-40,118-220
}
run sq_insts SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVES
run sq_time SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
run tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run ta TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
