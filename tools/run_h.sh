mkdir -p gpurun_out/r4
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "t_segments" > gpurun_out/r4/h3_ops.log 2>&1
tail -3 gpurun_out/r4/h3_ops.log
timeout -k 10 900 python tools/option_sweep.py 30 dw_tquad_wgs_fwd=1024 dw_tquad_wgs_fwd=2048 dw_tquad_wgs=1024 dw_tquad_wgs=1024,dw_tsplit_wgs=512 dw_tquad_wgs_fwd=1024,dw_tquad_wgs=1024,dw_tsplit_wgs=512 > gpurun_out/r4/h3_sweep.txt 2>&1
cat gpurun_out/r4/h3_sweep.txt
