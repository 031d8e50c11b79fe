mkdir -p gpurun_out/r4
timeout -k 10 300 python tools/mb_tiny.py > gpurun_out/r4/g_tiny.txt 2>&1
cat gpurun_out/r4/g_tiny.txt
