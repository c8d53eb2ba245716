cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "vit or token0 or g5 or aux or clip_forward" > gpurun_out/r6_tests_e.txt 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r6_tests_e.txt | cut -c1-300
for v in 2048 0 2048 0; do TRIS_GEMM_CONVERT=$v timeout 400 python bench.py --backbone clip-ViT-B/16 --steps 6 --warmup 2 --headline-only 2>/dev/null | sed "s/^/vitb16 convert=$v /"; done > gpurun_out/r6_convert_ab.txt
for v in 2048 0 2048 0; do TRIS_GEMM_CONVERT=$v timeout 300 python bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | sed "s/^/rn50 convert=$v /"; done >> gpurun_out/r6_convert_ab.txt
cat gpurun_out/r6_convert_ab.txt
