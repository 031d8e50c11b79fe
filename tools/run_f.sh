set -x
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "se_bwd or se_bn" > gpurun_out/r4/f_ops.log 2>&1 && \
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "blocks or 2x4x158 or 8x4x64" > gpurun_out/r4/f_model.log 2>&1
tail -3 gpurun_out/r4/f_ops.log; tail -3 gpurun_out/r4/f_model.log
bash tools/run_e.sh
