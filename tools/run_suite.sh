# the driver's GPU suite (round end rehearsal):  bash tools/run_suite.sh <tag> [ENV=VALUE ...]
TAG=${1:-s}; shift
mkdir -p gpurun_out/r4
env "$@" timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=8 > gpurun_out/r4/${TAG}_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r4/${TAG}_suite.log
tail -6 gpurun_out/r4/${TAG}_suite.log
