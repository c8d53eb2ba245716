cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 20 12 20 12; do TRIS_BENCH_QL=$v timeout 300 python bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | sed "s/^/query_len=$v /"; done > gpurun_out/r6_whatif_query_len.txt; cat gpurun_out/r6_whatif_query_len.txt
