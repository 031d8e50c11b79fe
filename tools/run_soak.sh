mkdir -p gpurun_out/r4
timeout -k 10 900 python tools/soak.py > gpurun_out/r4/soak.txt 2>&1
tail -15 gpurun_out/r4/soak.txt
