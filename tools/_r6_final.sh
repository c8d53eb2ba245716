cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r6_gputests_final.log 2>&1 < /dev/null; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/r6_gputests_final.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6_smoke_final.log 2>&1 < /dev/null; tail -1 gpurun_out/r6_smoke_final.log
TRIS_TUNE_LOG=gpurun_out/r6_autotune_log_all.txt timeout 1200 python bench.py > gpurun_out/r6_bench_final.log 2>&1 < /dev/null; grep "^{" gpurun_out/r6_bench_final.log > gpurun_out/r6_bench_final.json; cut -c1-300 gpurun_out/r6_bench_final.json
