# same-box A/B of two library builds: the tree (A) against the copy under _ab/b (B), alternating
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/ab.txt
: > $OUT
for i in 1 2; do
  echo "== A (tree) run $i" >> $OUT; timeout -k 10 200 python tools/option_sweep.py 30 >> $OUT 2>&1
  echo "== B (_ab/b) run $i" >> $OUT; timeout -k 10 200 python _ab/b/tools/option_sweep.py 30 >> $OUT 2>&1
done
grep -E "==|defaults" $OUT
