set -x
mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused" > gpurun_out/r4/c_ops.log 2>&1 && \
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_mixed_storage_gpu.py -x -q -m gpu -k "blocks or block_bf16 or 2x4x158 or 8x4x64 or 8x16x224 or kernel or poison" > gpurun_out/r4/c_model.log 2>&1
tail -3 gpurun_out/r4/c_ops.log; tail -3 gpurun_out/r4/c_model.log
bash tools/chain_r4.sh d
