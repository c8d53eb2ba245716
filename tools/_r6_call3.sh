cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_planes.py -q > gpurun_out/r6_tests_c.txt 2>&1; echo "planes tests rc $?"
for v in 1 0 1 0; do TRIS_AUX_TEXT_EARLY=$v timeout 300 python bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | sed "s/^/aux_early=$v /"; done > gpurun_out/r6_aux_early_ab.txt; cat gpurun_out/r6_aux_early_ab.txt
timeout 300 python tools/step_graph_marks.py > gpurun_out/r6_step_graph_marks.txt 2>&1; echo "marks rc $?"
TRIS_AUX_TEXT_EARLY=0 timeout 300 python tools/step_graph_marks.py > gpurun_out/r6_step_graph_marks_auxlate.txt 2>&1
( for env in "X=1" "TRIS_GEMM_MODE=x3" "TRIS_H2_PLANES=0" "TRIS_HIPGRAPH=0"; do echo "== $env"; env $env timeout 200 python tools/eval_throughput.py 2>/dev/null; done ) > gpurun_out/r6_eval_bisect.txt 2>&1; echo "eval rc $?"
( for own in 1 0; do echo "== own_stream=$own"; timeout 200 python - <<PY
import os, sys, json, torch
os.environ.setdefault("TRIS_RANDOM_INIT", "1"); sys.path.insert(0, ".")
from tris_amd import ops
if $own: torch.cuda.set_stream(ops.compute_stream())
from tools.eval_throughput import measure
print(json.dumps(measure()))
PY
done ) >> gpurun_out/r6_eval_bisect.txt 2>&1
# PMC traffic ledger: two passes of the instrumented step under rocprofv3
export TRIS_AUTOTUNE=0
LEDGER_ORDER=gpurun_out/r6_ledger_order_f.txt timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r6_lf -o f -- python tools/step_ledger.py 48 > gpurun_out/r6_ledger_fetch.log 2>&1; echo "fetch rc $?"
LEDGER_ORDER=gpurun_out/r6_ledger_order_w.txt timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r6_lw -o w -- python tools/step_ledger.py 48 > gpurun_out/r6_ledger_write.log 2>&1; echo "write rc $?"
F=$(ls gpurun_out/r6_lf/*/*counter_collection.csv gpurun_out/r6_lf/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls gpurun_out/r6_lw/*/*counter_collection.csv gpurun_out/r6_lw/*counter_collection.csv 2>/dev/null | head -1)
echo "F=$F W=$W"; head -2 $F
python tools/pmc_ledger.py gpurun_out/r6_ledger_order_f.txt $F $W gpurun_out/r6_traffic_ledger.txt > gpurun_out/r6_traffic_ledger.log 2>&1; echo "ledger rc $?"; head -12 gpurun_out/r6_traffic_ledger.log
rm -rf gpurun_out/r6_lf gpurun_out/r6_lw
unset TRIS_AUTOTUNE
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/r6_sq_counters.txt; wc -l gpurun_out/r6_sq_counters.txt
tail -4 gpurun_out/r6_tests_c.txt; cat gpurun_out/r6_step_graph_marks.txt | head -20
