TAG=${1:-p}
mkdir -p gpurun_out/r4/prof
bash tools/profile_round4.sh $TAG > gpurun_out/r4/prof_$TAG.log 2>&1
ls gpurun_out/r4/prof/
python3 bench.py > gpurun_out/r4/prof/${TAG}_bench_default.json 2> gpurun_out/r4/prof/${TAG}_bench_default.err
tail -c 900 gpurun_out/r4/prof/${TAG}_bench_default.json
