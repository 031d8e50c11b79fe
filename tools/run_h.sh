mkdir -p gpurun_out/r4
timeout -k 10 900 python tools/option_sweep.py 30 cfg.wgrad_overlap=1 cfg.wgrad_overlap=1,wg_cpw=2,wg_cap=1024 cfg.wgrad_overlap=1,wg_cpw=1,wg_cap=2048 cfg.wgrad_overlap=1,wg_cpw=4,wg_cap=512 wg_cpw=2,wg_cap=1024 cfg.side_stream=1 > gpurun_out/r4/h_sweep.txt 2>&1
cat gpurun_out/r4/h_sweep.txt
