cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 1 0; do TRIS_BN_BITMASK=$v timeout 300 python tools/step_ledger.py 48 gpurun_out/r6_ledger_bitmask$v.txt > /dev/null 2>&1; grep "gemm_bnbwd" gpurun_out/r6_ledger_bitmask$v.txt | awk -v v=$v '{s+=$1} END {print "bitmask=" v " gemm_bnbwd total ms " s}'; head -1 gpurun_out/r6_ledger_bitmask$v.txt; done
