# profiles/r02/d_*: the final build of round 2 in both storage modes (run on the MI355X box from the repo root)
set -x
mkdir -p gpurun_out/r2/profd && cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r2/profd
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 $R/bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16_M.json 2> $O/bench_bf16_M.err
python3 $R/bench.py --no-cpu-baseline --dtype bf16 --model L --size 312 --steps 10 > $O/bench_bf16_L312.json 2> $O/bench_bf16_L312.err
python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 --model L --size 312 --steps 10 > $O/bench_f32_L312.json 2> $O/bench_f32_L312.err
rocprofv3 --kernel-trace --stats -d $O/kt_f32 -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 > $O/bench_f32_under_rocprof.json 2> $O/kt_f32.err
rocprofv3 --kernel-trace --stats -d $O/kt_bf16 -- python3 $R/bench.py --no-cpu-baseline --dtype bf16 --model L --size 312 --steps 10 > $O/bench_bf16_L312_under_rocprof.json 2> $O/kt_bf16.err
B="python3 $R/bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-exact-fp32 --dtype bf16"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- $B > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- $B > $O/pmc_w.log 2>&1
cd $R
python tools/profile_summary.py $(ls $O/kt_f32/*/*.db | head -1) $O/kernel_stats_f32.csv > $O/fam_f32.txt 2>&1
python tools/profile_summary.py $(ls $O/kt_bf16/*/*.db | head -1) $O/kernel_stats_bf16_L312.csv > $O/fam_bf16_L312.txt 2>&1
python tools/collect_traffic.py $O/pmc_fetch $O/pmc_write $O/traffic_bf16_M.json > $O/traffic_bf16_M.txt 2>&1
python tools/dw_scaling.py > $O/dw_scaling.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/kt_f32 $O/kt_bf16
echo done
