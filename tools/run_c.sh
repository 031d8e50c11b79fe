set -x
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "sixteen or test_pw_fwd or test_pw_bwd" > gpurun_out/r4/c_ops.log 2>&1 && \
timeout -k 10 300 python tools/mb_pw8.py > gpurun_out/r4/c_mb.txt 2>&1
tail -5 gpurun_out/r4/c_ops.log; cat gpurun_out/r4/c_mb.txt
