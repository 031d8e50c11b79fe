# Final collection of round 3 (run from the repo root through gpurun): the default bench line, the rocprofv3 kernel trace of
# the same command, the PMC passes (tools/profile_round3.sh), the multigrid sweep and the other storage / model lines.
set -x
TAG=${1:-d}
R=$(pwd); OUT=$R/gpurun_out/r3/prof; mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
echo "bench done"
bash tools/profile_round3.sh $TAG
python tools/multigrid_sweep.py 8 > $OUT/${TAG}_multigrid_shapes.txt 2>&1
python3 bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timing > $OUT/${TAG}_bench_bf16_M.json 2>/dev/null
python3 bench.py --model L --size 312 --no-cpu-baseline --no-kernel-timing --no-exact-fp32 > $OUT/${TAG}_bench_f32_L312.json 2>/dev/null
python3 bench.py --model L --size 312 --dtype bf16 --no-cpu-baseline --no-kernel-timing > $OUT/${TAG}_bench_bf16_L312.json 2>/dev/null
echo all done
