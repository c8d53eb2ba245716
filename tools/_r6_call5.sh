cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2; do for v in unset 2 1; do
  if [ $v = unset ]; then unset TRIS_XCD_ORDER; else export TRIS_XCD_ORDER=$v; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | sed "s/^/xcd_order=$v /"
done; done > gpurun_out/r6_xcd_order_ab.txt; cat gpurun_out/r6_xcd_order_ab.txt
