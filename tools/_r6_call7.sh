cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6_gputests_a.txt 2>&1; echo "gpu tests rc $?"; tail -5 gpurun_out/r6_gputests_a.txt
timeout 900 python bench.py > gpurun_out/r6_bench_full_a.json 2> gpurun_out/r6_bench_full_a.err; echo "bench rc $?"; tail -3 gpurun_out/r6_bench_full_a.err; head -c 400 gpurun_out/r6_bench_full_a.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/r6_smoke.log
