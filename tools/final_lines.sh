R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R
make -C tests/fake_rccl > /dev/null 2>&1
echo "== c4 driver style"; timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_c4_bench_final.json 2> $OUT/r03_c4_bench_final.err; echo rc=$?
for c in c1 c2 c3 c5; do echo "== $c"; timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 > $OUT/r03_${c}_bench.json 2> $OUT/r03_${c}_bench.err; echo rc=$?; done
echo "== slab rank probe"
export MG_RCCL_LIBRARY=$R/tests/fake_rccl/libfake_rccl_loopback.so
rm -f $OUT/r03_slab_rank_final.txt
for mu in 50 2; do for t in "" halo_depth=5 halo_depth=5,overlap=0; do timeout -k 10 300 python tools/slab_rank_probe.py 8 7 $mu "$t" >> $OUT/r03_slab_rank_final.txt 2>&1; tail -1 $OUT/r03_slab_rank_final.txt; done; done
export MG_RCCL_LIBRARY=$R/tests/fake_rccl/libfake_rccl.so
timeout -k 10 300 python tools/slab_overhead_probe.py 8 7 50 > $OUT/r03_slab8.txt 2>&1; tail -2 $OUT/r03_slab8.txt
echo "== done"
