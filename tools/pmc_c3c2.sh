R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=r03
export TMPDIR=/tmp; cd /tmp
for CFG in c3 c2; do
  for C in FETCH_SIZE WRITE_SIZE; do
    N=$(echo $C | tr 'A-Z' 'a-z' | sed 's/_size//')
    echo "pmc $CFG $N"
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/${TAG}_${CFG}_pmc_$N.d" -- python3 "$R/bench.py" --config $CFG --steps 2 --warmup 1 --kernel-reps 4 --no-cpu-baseline > /dev/null 2> "$OUT/${TAG}_${CFG}_pmc_$N.err"
    python3 "$R/profiles/summarize.py" pmc "$OUT/${TAG}_${CFG}_pmc_$N.d" "$OUT/${TAG}_${CFG}_pmc_$N.csv"
  done
done
rm -rf "$OUT"/${TAG}_c3_pmc_*.d "$OUT"/${TAG}_c2_pmc_*.d
echo done
