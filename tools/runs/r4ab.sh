#!/bin/bash
# the vector memory pipeline of k_samples_lean (TA / TCP counters, two per block and pass): is it what the kernel waits for?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r4ab
L="--pipeline-seconds 0 --e2e-seconds 0"
{
for set in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum"; do
  echo "== $set"
  PMC_TIMEOUT=150 bash tools/pmc_quick.sh "$set" $L | grep "k_samples_lean\|k_part_events<0\|k_part_hand_c\|k_part_hist"
  grep -h "error code" gpurun_out/pmcq/q.log | head -1
done
} > gpurun_out/r4ab/pmc.log 2>&1
cut -c1-400 gpurun_out/r4ab/pmc.log
