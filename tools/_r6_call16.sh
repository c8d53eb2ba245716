cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_planes.py -q -x > gpurun_out/r6_tests_j.txt 2>&1; echo "planes rc $?"; tail -4 gpurun_out/r6_tests_j.txt | cut -c1-250
for v in 1 0 1 0; do TRIS_BN_BITMASK=$v timeout 300 python bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | sed "s/^/bn_bitmask=$v /"; done > gpurun_out/r6_bn_bitmask_ab.txt; cat gpurun_out/r6_bn_bitmask_ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step_graph.py -q -x -k "g5 or headline_batch_48 or replayed_step_equals" > gpurun_out/r6_tests_k.txt 2>&1; echo "parity rc $?"; tail -3 gpurun_out/r6_tests_k.txt | cut -c1-250
