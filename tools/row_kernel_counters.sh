#!/bin/bash
# SQ / TCP counters of the row kernel and the long-row kernel of the K loop at C = 128 on the config-4
# graph, side by side (same gather shape: what does the row kernel spend that the long-row kernel does not?)
export TMPDIR=/tmp
# (the four TA counters together exceed what one pass can collect on gfx950: two per pass)
PMC_FILTER=k_spmm bash tools/pmc_passes.sh r4f/c${1:-128} "tools/narrow_order_experiment.py --only workload --feats ${1:-128} --rounds 2" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
  "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum" \
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum SQ_INST_CYCLES_VMEM_RD"
