# same-box A/B of two library builds: the tree (A) against the copy under _ab/b (B), alternating; kernel-level tests of A first
mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "test_pw_bwd or sixteen or three_term or se_bwd" > gpurun_out/r4/ab_ops.log 2>&1
tail -2 gpurun_out/r4/ab_ops.log
OUT=gpurun_out/r4/ab.txt
: > $OUT
for i in 1 2; do
  echo "== A (tree) run $i" >> $OUT; timeout -k 10 200 python tools/option_sweep.py 30 >> $OUT 2>&1
  echo "== B (_ab/b) run $i" >> $OUT; timeout -k 10 200 python _ab/b/tools/option_sweep.py 30 >> $OUT 2>&1
done
grep -E "==|defaults" $OUT
