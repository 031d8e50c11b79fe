mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "test_pw_fwd" > gpurun_out/r4/h4_ops.log 2>&1
tail -2 gpurun_out/r4/h4_ops.log
timeout -k 10 600 python tools/option_sweep.py 30 pwfs_contract=1 > gpurun_out/r4/h4_sweep.txt 2>&1
cat gpurun_out/r4/h4_sweep.txt
