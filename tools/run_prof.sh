TAG=${1:-p}
mkdir -p gpurun_out/r4/prof
bash tools/profile_round4.sh $TAG > gpurun_out/r4/prof_$TAG.log 2>&1
ls gpurun_out/r4/prof/
bash tools/chain_r4.sh ${TAG}
python tools/multigrid_sweep.py 8 > gpurun_out/r4/prof/${TAG}_multigrid_shapes.txt 2>&1
python3 bench.py --dtype bf16 --no-cpu-baseline --no-kernel-timing > gpurun_out/r4/prof/${TAG}_bench_bf16_M.json 2>/dev/null
python3 bench.py --model L --size 312 --no-cpu-baseline --no-kernel-timing --no-exact-fp32 > gpurun_out/r4/prof/${TAG}_bench_f32_L312.json 2>/dev/null
python3 bench.py --model L --size 312 --dtype bf16 --no-cpu-baseline --no-kernel-timing > gpurun_out/r4/prof/${TAG}_bench_bf16_L312.json 2>/dev/null
echo all done
