#!/bin/bash
# Reproduces the rocprofv3 passes summarised under profiles/ (run on the GPU box, e.g. through gpurun):
#     bash tools/profile_round.sh r02
# kernel trace + stats of the default bench run, then separate PMC passes (FETCH_SIZE, WRITE_SIZE, L2) of a
# short run.  rocprofv3 must launch python3 itself (no env/bash wrapper between it and the program).
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_c4_trace" -- python3 "$R/bench.py" \
    > "$OUT/${TAG}_c4_bench.json" 2> "$OUT/${TAG}_c4_bench.err"
SHORT="--steps 1 --warmup 0 --mu 1 --kernel-reps 2 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_c4_pmc_fetch" -- python3 "$R/bench.py" $SHORT > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/${TAG}_c4_pmc_write" -- python3 "$R/bench.py" $SHORT > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/${TAG}_c4_pmc_l2" -- python3 "$R/bench.py" $SHORT > /dev/null 2>&1
cd "$R"
python3 bench.py > "$OUT/${TAG}_c4_bench_plain.json" 2> "$OUT/${TAG}_c4_plain.err"
echo "done: summarise with  python profiles/summarize.py trace|pmc gpurun_out/${TAG}_c4_... profiles/${TAG}_..."
