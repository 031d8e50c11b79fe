mkdir -p gpurun_out/r4/prof
S=$(date +%s)
python3 bench.py > gpurun_out/r4/prof/p3_bench_default.json 2> gpurun_out/r4/prof/p3_bench_default.err
echo "bench wall seconds: $(( $(date +%s) - S ))"
tail -c 700 gpurun_out/r4/prof/p3_bench_default.json
