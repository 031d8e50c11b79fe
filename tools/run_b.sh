set -x
mkdir -p gpurun_out/r4
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "se_bn or elementwise" > gpurun_out/r4/b_ops.log 2>&1 && \
python -m pytest tests/test_model_gpu.py -x -q -m gpu -s > gpurun_out/r4/b_model.log 2>&1 && \
python3 bench.py --no-cpu-baseline --no-exact-fp32 --no-kernel-timing > gpurun_out/r4/b_bench.json 2> gpurun_out/r4/b_bench.err
tail -3 gpurun_out/r4/b_ops.log; tail -5 gpurun_out/r4/b_model.log; cut -c1-300 gpurun_out/r4/b_bench.json
