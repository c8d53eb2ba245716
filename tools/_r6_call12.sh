cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "packed_text or g5 or g1_text or three_step or headline_batch_48" > gpurun_out/r6_tests_f.txt 2>&1; echo "tests rc $?"; tail -6 gpurun_out/r6_tests_f.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_ddp.py -q -x > gpurun_out/r6_tests_g.txt 2>&1; echo "tests2 rc $?"; tail -4 gpurun_out/r6_tests_g.txt | cut -c1-250
