cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "three_step" 2>&1 | grep -E "trajectory|passed|failed" > gpurun_out/r6_trajectory.txt; cat gpurun_out/r6_trajectory.txt
timeout 1500 python -m pytest tests/test_gpu_planes.py tests/test_gpu_step_graph.py tests/test_gpu_ddp.py -q -x > gpurun_out/r6_tests_h.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r6_tests_h.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "g5 or headline_batch_48 or g2 or g4" > gpurun_out/r6_tests_i.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r6_tests_i.txt | cut -c1-200
