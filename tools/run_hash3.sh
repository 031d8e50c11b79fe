mkdir -p gpurun_out/r4
timeout -k 10 300 python tests/golden/make_grad_hashes.py gpurun_out/r4/f3_grad_hashes.json > gpurun_out/r4/f3_hashes.log 2>&1
tail -9 gpurun_out/r4/f3_hashes.log
timeout -k 10 300 python -m pytest tests/test_determinism_gpu.py -q -m gpu 2>&1 | tail -2
