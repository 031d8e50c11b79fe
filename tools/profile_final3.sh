# final bench lines of round 3 + rocprofv3 kernel trace of the same command (kernel sources unchanged since the a_* PMC passes)
set -x
R=$(pwd); OUT=$R/gpurun_out/r3/prof; mkdir -p $OUT
python3 bench.py > $OUT/b_bench_default.json 2> $OUT/b_bench_default.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 > $OUT/b_bench_under_rocprof.json 2> $OUT/kt.err
echo "kernel trace done"
cd $R
python tools/profile_summary.py $(ls $OUT/kt/*/*.db | head -1) $OUT/b_kernel_stats.csv > $OUT/b_families.txt 2>&1
rm -rf $OUT/kt
python tools/multigrid_sweep.py 8 > $OUT/b_multigrid_shapes.txt 2>&1
echo done
