# Round-end rehearsal on one box: gradient hashes of the current kernel sources, the whole GPU suite (pinned parity), smoke.
#   bash tools/run_final.sh <tag>
TAG=${1:-f}
mkdir -p gpurun_out/r4
timeout -k 10 300 python tests/golden/make_grad_hashes.py gpurun_out/r4/${TAG}_grad_hashes.json > gpurun_out/r4/${TAG}_hashes.log 2>&1
cat gpurun_out/r4/${TAG}_hashes.log | tail -9
cp gpurun_out/r4/${TAG}_grad_hashes.json tests/golden/grad_hashes.json
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=6 > gpurun_out/r4/${TAG}_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r4/${TAG}_suite.log
tail -12 gpurun_out/r4/${TAG}_suite.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/r4/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r4/${TAG}_smoke.log
