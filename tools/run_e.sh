# dependent-launch cost of the replayed graph under runtime environment knobs (round 4)
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/e_launch_env.txt
: > $OUT
run() {
  echo "=== $*" >> $OUT
  env "$@" timeout -k 10 120 python tools/launch_chain.py >> $OUT 2>&1
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-fp32 --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], 'clips/s', d['ms_per_step'], 'ms')" >> $OUT 2>&1
}
run X3D_DUMMY=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run GPU_MAX_HW_QUEUES=1
run ROC_ACTIVE_WAIT_TIMEOUT=0
cat $OUT
