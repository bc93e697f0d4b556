#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 passes over tools/pmc_jacobi2.py (run on the GPU box):  bash tools/pmc_jacobi2.sh TAG [key=value ...]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out
mkdir -p "$OUT"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_$N" -- python3 "$R/tools/pmc_jacobi2.py" "$@" > "$OUT/${TAG}_$N.log" 2>&1
  python3 "$R/profiles/summarize.py" pmc "$OUT/${TAG}_$N" "$OUT/${TAG}_$N.csv"
  grep -i "jacobi2" "$OUT/${TAG}_$N.csv" | cut -c1-40,100-
done
