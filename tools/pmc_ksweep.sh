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
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_sq$i" -- python3 "$R/tools/pmc_ksweep.py" "$@" > "$OUT/${TAG}_sq$i.log" 2>&1 || { echo "group $i failed"; tail -3 "$OUT/${TAG}_sq$i.log"; continue; }
  python3 "$R/profiles/summarize.py" pmc "$OUT/${TAG}_sq$i" "$OUT/${TAG}_sq$i.csv"
  grep -i "jacobikc" "$OUT/${TAG}_sq$i.csv" | sed 's/"void mgk::sdia_jacobikc[^"]*"/JKC/'
done
