#!/bin/bash
# Reproduces the rocprofv3 passes summarised under profiles/ (run on the GPU box, e.g. through gpurun):
#     bash tools/profile_round.sh r02
# 1. kernel trace + stats of the default bench run (the headline workload),
# 2. separate PMC passes (FETCH_SIZE, WRITE_SIZE, L2 hit/miss) of a short bench run -- counters never share a run
#    with --stats or another trace domain --,
# 3. SQ wait / busy / instruction counters of the K-sweep pass alone (tools/pmc_ksweep.py),
# each condensed with profiles/summarize.py into gpurun_out/${TAG}_*.csv (copy those into profiles/).
# rocprofv3 must launch python3 itself (no env/bash wrapper between it and the program).
set -e
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out
mkdir -p "$OUT"
echo "trace"; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_c4_trace" -- python3 "$R/bench.py" \
    > "$OUT/${TAG}_c4_bench_under_rocprof.json" 2> "$OUT/${TAG}_c4_bench.err"
python3 "$R/profiles/summarize.py" trace "$OUT/${TAG}_c4_trace" "$OUT/${TAG}_c4_kernel_by_grid.csv"
cp "$(ls "$OUT/${TAG}_c4_trace"/*/*kernel_stats.csv | head -1)" "$OUT/${TAG}_c4_kernel_stats.csv"
SHORT="--steps 1 --warmup 0 --mu 1 --kernel-reps 2 --no-cpu-baseline --no-odd-rows"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  N=$(echo $C | cut -d' ' -f1 | tr 'A-Z' 'a-z' | sed 's/_size//; s/tcc_hit_sum/l2/')
  echo "pmc $N"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_c4_pmc_$N.d" -- python3 "$R/bench.py" $SHORT > /dev/null 2> "$OUT/${TAG}_c4_pmc_$N.err"
  python3 "$R/profiles/summarize.py" pmc "$OUT/${TAG}_c4_pmc_$N.d" "$OUT/${TAG}_c4_pmc_$N.csv"
done
for C in FETCH_SIZE WRITE_SIZE; do
  N=$(echo $C | tr 'A-Z' 'a-z' | sed 's/_size//')
  echo "pmc c5 $N"
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_c5_pmc_$N.d" -- python3 "$R/bench.py" --config c5 $SHORT > /dev/null 2> "$OUT/${TAG}_c5_pmc_$N.err"
  python3 "$R/profiles/summarize.py" pmc "$OUT/${TAG}_c5_pmc_$N.d" "$OUT/${TAG}_c5_pmc_$N.csv"
done
# the other configurations whose dominant launch has a traffic entry: C3 (four-sweep march on 257^3 rows), C2 (2-D K-sweep kernel)
for CFG in c3 c2; do
  for C in FETCH_SIZE WRITE_SIZE; do
    N=$(echo $C | tr 'A-Z' 'a-z' | sed 's/_size//')
    echo "pmc $CFG $N"
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_${CFG}_pmc_$N.d" -- python3 "$R/bench.py" --config $CFG --steps 2 --warmup 1 --kernel-reps 4 --no-cpu-baseline > /dev/null 2> "$OUT/${TAG}_${CFG}_pmc_$N.err"
    python3 "$R/profiles/summarize.py" pmc "$OUT/${TAG}_${CFG}_pmc_$N.d" "$OUT/${TAG}_${CFG}_pmc_$N.csv"
  done
done
echo "sq"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/${TAG}_c4_pmc_sq.d" -- python3 "$R/tools/pmc_ksweep.py" > "$OUT/${TAG}_c4_pmc_sq.log" 2>&1
python3 "$R/profiles/summarize.py" pmc "$OUT/${TAG}_c4_pmc_sq.d" "$OUT/${TAG}_c4_pmc_sq.csv"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$OUT/${TAG}_c4_pmc_sq2.d" -- python3 "$R/tools/pmc_ksweep.py" > "$OUT/${TAG}_c4_pmc_sq2.log" 2>&1
python3 "$R/profiles/summarize.py" pmc "$OUT/${TAG}_c4_pmc_sq2.d" "$OUT/${TAG}_c4_pmc_sq2.csv"
cd "$R"
echo "plain"; python3 bench.py > "$OUT/${TAG}_c4_bench_untraced.json" 2> "$OUT/${TAG}_c4_untraced.err"
rm -rf "$OUT/${TAG}_c4_trace" "$OUT"/${TAG}_c4_pmc_*.d "$OUT"/${TAG}_c5_pmc_*.d "$OUT"/${TAG}_c3_pmc_*.d "$OUT"/${TAG}_c2_pmc_*.d
echo "done"
