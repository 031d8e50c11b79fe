# rocprofv3 kernel trace + per-template summary of the default bench command (run from the repo root through gpurun):
#   bash tools/kernel_trace.sh <tag>   ->  gpurun_out/r3/prof/<tag>_{kernel_stats.csv,families.txt,bench_under_rocprof.json}
TAG=${1:-k}
R=$(pwd); OUT=$R/gpurun_out/r3/prof; mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/kt.err
cd $R
python tools/profile_summary.py $(ls $OUT/kt/*/*.db | head -1) $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_families.txt 2>&1
rm -rf $OUT/kt
head -12 $OUT/${TAG}_families.txt | cut -c1-100 | grep -v Cijk
