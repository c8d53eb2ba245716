cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python tools/planes_bench.py vit > gpurun_out/r6_vit_planes_bench.txt 2>&1; echo "rc $?"; grep -v amdgpu gpurun_out/r6_vit_planes_bench.txt | cut -c1-200
