#!/bin/bash
# SQ / LDS counters of the K-sweep pass (tools/pmc_ksweep.py), one pass per counter group.  bash tools/pmc_ksweep.sh TAG key=value ...
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out
mkdir -p "$OUT"
i=0
GROUPS_FROM=${MG_PMC_FROM:-1}
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  [ $i -lt $GROUPS_FROM ] && continue
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_sq$i" -- python3 "$R/tools/pmc_ksweep.py" "$@" > "$OUT/${TAG}_sq$i.log" 2>&1 || { echo "group $i failed"; tail -3 "$OUT/${TAG}_sq$i.log"; continue; }
  python3 "$R/profiles/summarize.py" pmc "$OUT/${TAG}_sq$i" "$OUT/${TAG}_sq$i.csv"
  grep -i "jacobikc" "$OUT/${TAG}_sq$i.csv" | sed 's/"void mgk::sdia_jacobikc[^"]*"/JKC/'
done
