set -x
mkdir -p gpurun_out/r2/prof && cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 -L > $R/gpurun_out/r2/prof/counters.txt 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2/prof/kt -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 > $R/gpurun_out/r2/prof/bench_under_rocprof.json 2> $R/gpurun_out/r2/prof/kt.err
B="python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-exact-fp32"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r2/prof/pmc_fetch -- $B > $R/gpurun_out/r2/prof/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r2/prof/pmc_write -- $B > $R/gpurun_out/r2/prof/pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $R/gpurun_out/r2/prof/pmc_lds -- $B > $R/gpurun_out/r2/prof/pmc_l.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -d $R/gpurun_out/r2/prof/pmc_mfma -- $B > $R/gpurun_out/r2/prof/pmc_m.log 2>&1
cd $R
python tools/profile_summary.py $(ls gpurun_out/r2/prof/kt/*/*.db | head -1) gpurun_out/r2/prof/kernel_stats.csv > gpurun_out/r2/prof/fam.txt 2>&1
python tools/collect_traffic.py gpurun_out/r2/prof/pmc_fetch gpurun_out/r2/prof/pmc_write gpurun_out/r2/prof/traffic.json > gpurun_out/r2/prof/traffic.txt 2>&1
python tools/collect_pmc.py gpurun_out/r2/prof/pmc_lds gpurun_out/r2/prof/pmc_mfma gpurun_out/r2/prof/pmc_sq.json > gpurun_out/r2/prof/pmc_sq.txt 2>&1
rm -rf gpurun_out/r2/prof/pmc_fetch gpurun_out/r2/prof/pmc_write gpurun_out/r2/prof/pmc_lds gpurun_out/r2/prof/pmc_mfma gpurun_out/r2/prof/kt
echo done
