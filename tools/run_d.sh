set -x
mkdir -p gpurun_out/r4
timeout -k 10 900 python tools/option_sweep.py 30 pw8_max_k=0 pw8_max_k=224 pw8_max_k=224,pw9_max_k=224 pw9_max_k=128 fb_grid=256 fb_grid=384 wg_cpw=16 wg_cap=128 > gpurun_out/r4/d_sweep.txt 2>&1
cat gpurun_out/r4/d_sweep.txt
