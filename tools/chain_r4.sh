# Round 4: ordered kernel timeline of one replayed step (tools/chain_trace.py) + the HIP-event launch table of bench.py.
#   bash tools/chain_r4.sh <tag>      (through gpurun, from the repo root)
set -x
TAG=${1:-a}
R=$(pwd); OUT=$R/gpurun_out/r4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace -d /tmp/kt_$TAG -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 --no-kernel-timing --steps 6 > $OUT/${TAG}_bench_under_trace.json 2> $OUT/${TAG}_trace.err
cd $R
python3 tools/chain_trace.py /tmp/kt_$TAG $OUT/${TAG}_chain.txt
python3 bench.py --no-cpu-baseline --dump-launches $OUT/${TAG}_launches.json > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
