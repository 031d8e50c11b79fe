# Round-4 evidence collection on the GPU box (run from the repo root through gpurun): rocprofv3 kernel trace + stats of the
# default bench command, then separate --pmc passes (HBM traffic: FETCH_SIZE, WRITE_SIZE; SQ counters) of the eager step.
# Every pass carries --kernel-trace only, as the pool requires.  Output: gpurun_out/r4/prof/ (copied to profiles/r04/ by hand).
set -x
TAG=${1:-a}
R=$(pwd)
OUT=$R/gpurun_out/r4/prof
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/kt.err
echo "kernel trace done"
B="python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-exact-fp32"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -- $B > $OUT/pmc_f.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -- $B > $OUT/pmc_w.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/pmc_lds -- $B > $OUT/pmc_l.log 2>&1
echo "lds pass done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -d $OUT/pmc_mfma -- $B > $OUT/pmc_m.log 2>&1
echo "mfma pass done"
cd $R
python tools/profile_summary.py $(ls $OUT/kt/*/*.db | head -1) $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_families.txt 2>&1
python tools/collect_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_traffic_f32_M.json > $OUT/${TAG}_traffic_f32_M.txt 2>&1
python tools/collect_pmc.py $OUT/pmc_lds $OUT/pmc_mfma $OUT/${TAG}_pmc_sq_f32_M.json > $OUT/${TAG}_pmc_sq_f32_M.txt 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_lds $OUT/pmc_mfma $OUT/kt
echo done
