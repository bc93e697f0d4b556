#!/bin/bash
# Everything the round's summaries are made from, in one GPU call:  bash tools/final_round.sh r02
#   full GPU test suite, profile passes of the headline run (tools/profile_round.sh), bench lines of the other BASELINE
#   configurations, the one-GPU slab overhead probe.  Progress goes to gpurun_out/${TAG}_final.log.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
LOG=$OUT/${TAG}_final.log
cd "$R"
make -C tests/fake_rccl > /dev/null 2>&1
echo "== pytest" > "$LOG"
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > "$OUT/${TAG}_gpu_tests.log" 2>&1; echo "pytest rc=$?" >> "$LOG"; tail -3 "$OUT/${TAG}_gpu_tests.log" >> "$LOG"
echo "== profile" >> "$LOG"
timeout -k 10 600 bash tools/profile_round.sh "$TAG" >> "$LOG" 2>&1
for c in c1 c2 c3 c5; do
  echo "== bench $c" >> "$LOG"
  timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 > "$OUT/${TAG}_${c}_bench.json" 2> "$OUT/${TAG}_${c}_bench.err"; echo "rc=$?" >> "$LOG"
done
echo "== slab probe" >> "$LOG"
export MG_RCCL_LIBRARY=$R/tests/fake_rccl/libfake_rccl.so
timeout -k 10 300 python tools/slab_overhead_probe.py 8 7 50 > "$OUT/${TAG}_slab8.txt" 2>&1; tail -2 "$OUT/${TAG}_slab8.txt" >> "$LOG"
timeout -k 10 200 python tools/slab_overhead_probe.py 2 7 50 > "$OUT/${TAG}_slab2.txt" 2>&1; tail -1 "$OUT/${TAG}_slab2.txt" >> "$LOG"
echo "== one slab of eight alone (loopback stand-in)" >> "$LOG"
export MG_RCCL_LIBRARY=$R/tests/fake_rccl/libfake_rccl_loopback.so
for mu in 50 2; do
  for t in "" halo_depth=5 halo_depth=5,overlap=0; do
    timeout -k 10 300 python tools/slab_rank_probe.py 8 7 $mu "$t" >> "$OUT/${TAG}_slab_rank.txt" 2>&1; tail -1 "$OUT/${TAG}_slab_rank.txt" >> "$LOG"
  done
done
echo "== done" >> "$LOG"
cat "$LOG"
