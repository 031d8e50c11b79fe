mkdir -p gpurun_out/r2/prof2
python -m pytest tests/test_ops_gpu.py -x -q -k "dw333" > gpurun_out/r2/t13.log 2>&1; tail -3 gpurun_out/r2/t13.log
python bench.py --no-cpu-baseline --no-exact-fp32 --dump-launches gpurun_out/r2/launches13.json > gpurun_out/r2/b13.json 2> gpurun_out/r2/b13.err; cut -c1-260 gpurun_out/r2/b13.json
cd /tmp && export TMPDIR=/tmp
R=/root/repo
B="python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-exact-fp32"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $R/gpurun_out/r2/prof2/pmc_lds -- $B > $R/gpurun_out/r2/prof2/pmc_l.log 2>&1
cd $R
python tools/collect_pmc.py gpurun_out/r2/prof2/pmc_lds gpurun_out/r2/prof2/pmc_sq.json > gpurun_out/r2/prof2/pmc_sq.txt 2>&1
rm -rf gpurun_out/r2/prof2/pmc_lds
grep -A1 "^dw333" gpurun_out/r2/prof2/pmc_sq.txt
