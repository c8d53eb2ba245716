# closing run of a round (on the GPU box: gpurun -- bash tools/closing_run.sh): full GPU test suite, smoke, bench (with CPU baseline), then the profile set of tools/closing_profiles.sh; outputs under gpurun_out/, copied to profiles/ by hand
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export TAG=${TAG:-r3z}
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gputests.log 2>&1 < /dev/null; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_gputests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1 < /dev/null; tail -1 gpurun_out/${TAG}_smoke.log
TRIS_TUNE_LOG=gpurun_out/${TAG}_tune.txt timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1 < /dev/null; grep "^{" gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json
timeout 400 python bench.py --backbone clip-ViT-B/16 --no-cpu-baseline --no-pipeline > gpurun_out/${TAG}_bench_vit.log 2>&1 < /dev/null; grep "^{" gpurun_out/${TAG}_bench_vit.log > gpurun_out/${TAG}_bench_vit.json; cut -c1-200 gpurun_out/${TAG}_bench_vit.json
timeout 300 python tools/gemm_wp_bench.py > gpurun_out/${TAG}_gemm_weight_planes.txt 2>&1 < /dev/null; tail -3 gpurun_out/${TAG}_gemm_weight_planes.txt | cut -c1-160
timeout 300 python tools/step_graph_marks.py > gpurun_out/${TAG}_step_graph_marks.txt 2>&1 < /dev/null; tail -2 gpurun_out/${TAG}_step_graph_marks.txt
bash tools/closing_profiles.sh
