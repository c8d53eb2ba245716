cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "packed_text or g5 or g1_text or three_step" > gpurun_out/r6_tests_f.txt 2>&1; echo "tests rc $?"; tail -12 gpurun_out/r6_tests_f.txt | cut -c1-250
for v in 1 0 1 0; do TRIS_TEXT_PACK=$v timeout 300 python bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | sed "s/^/text_pack=$v /"; done > gpurun_out/r6_text_pack_ab.txt; cat gpurun_out/r6_text_pack_ab.txt
timeout 300 python tools/step_graph_marks.py 2>/dev/null | grep -E "text_fwd|aux_text|trunk_fwd|heads_fwd|b0_done|opt_done"
