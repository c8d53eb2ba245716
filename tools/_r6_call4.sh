cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "token0 or g7 or validate" > gpurun_out/r6_tests_d.txt 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r6_tests_d.txt
timeout 200 python tools/eval_throughput.py 2>/dev/null > gpurun_out/r6_eval_after.txt; echo "eval rc $?"; cut -c1-300 gpurun_out/r6_eval_after.txt
for v in 1 0 1 0; do TRIS_VIT_TOKEN0=$v timeout 300 python bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | sed "s/^/vit_token0=$v /"; done > gpurun_out/r6_token0_ab.txt; cat gpurun_out/r6_token0_ab.txt
# SQ counters of the probe kernels (4096^3 only), two passes
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_MFMA"
timeout 300 rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d gpurun_out/r6_pp1 -o a -- tools/probes/h2_phase_probe 1 1 > gpurun_out/r6_probe_pmc1.log 2>&1; echo "pmc1 rc $?"
timeout 300 rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d gpurun_out/r6_pp2 -o b -- tools/probes/h2_phase_probe 1 1 > gpurun_out/r6_probe_pmc2.log 2>&1; echo "pmc2 rc $?"
A=$(ls gpurun_out/r6_pp1/*counter_collection.csv gpurun_out/r6_pp1/*/*counter_collection.csv 2>/dev/null | head -1)
B=$(ls gpurun_out/r6_pp2/*counter_collection.csv gpurun_out/r6_pp2/*/*counter_collection.csv 2>/dev/null | head -1)
python tools/probe_pmc_summary.py gpurun_out/r6_phase_probe_pmc.txt $A $B > /dev/null; echo "summary rc $?"; head -30 gpurun_out/r6_phase_probe_pmc.txt
rm -rf gpurun_out/r6_pp1 gpurun_out/r6_pp2
# PMC traffic ledger (join fixed)
export TRIS_AUTOTUNE=0
LEDGER_ORDER=gpurun_out/r6_ledger_order_f.txt timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r6_lf -o f -- python tools/step_ledger.py 48 > gpurun_out/r6_ledger_fetch.log 2>&1; echo "fetch rc $?"
LEDGER_ORDER=gpurun_out/r6_ledger_order_w.txt timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r6_lw -o w -- python tools/step_ledger.py 48 > gpurun_out/r6_ledger_write.log 2>&1; echo "write rc $?"
F=$(ls gpurun_out/r6_lf/*counter_collection.csv gpurun_out/r6_lf/*/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls gpurun_out/r6_lw/*counter_collection.csv gpurun_out/r6_lw/*/*counter_collection.csv 2>/dev/null | head -1)
python - "$F" gpurun_out/r6_fetch_min.csv <<'PY'
import csv,sys
w=csv.writer(open(sys.argv[2],"w")); w.writerow(["Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"])
for r in csv.DictReader(open(sys.argv[1])): w.writerow([r["Dispatch_Id"],r["Kernel_Name"][:120],r["Counter_Name"],r["Counter_Value"]])
PY
python - "$W" gpurun_out/r6_write_min.csv <<'PY'
import csv,sys
w=csv.writer(open(sys.argv[2],"w")); w.writerow(["Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"])
for r in csv.DictReader(open(sys.argv[1])): w.writerow([r["Dispatch_Id"],r["Kernel_Name"][:120],r["Counter_Name"],r["Counter_Value"]])
PY
gzip -f gpurun_out/r6_fetch_min.csv gpurun_out/r6_write_min.csv
python tools/pmc_ledger.py gpurun_out/r6_ledger_order_f.txt $F $W gpurun_out/r6_traffic_ledger.txt > gpurun_out/r6_traffic_ledger.log 2>&1; echo "ledger rc $?"; head -14 gpurun_out/r6_traffic_ledger.log | cut -c1-170
rm -rf gpurun_out/r6_lf gpurun_out/r6_lw
