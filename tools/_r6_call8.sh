cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 1 0 1 0; do TRIS_FUSE_SPLITK_PY=$v timeout 300 python tools/step_ledger.py 48 gpurun_out/r6_ledger_fuse$v.txt 2>/dev/null | grep "^# tools\|time above" | sed "s/^/fuse=$v /"; done
