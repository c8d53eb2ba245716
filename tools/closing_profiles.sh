# measurement set of a round for ONE arithmetic (run on the GPU box through gpurun):
#   TAG=r5 MODE=h2 bash tools/closing_profiles.sh      -> gpurun_out/${TAG}_${MODE}_*   (MODE = h2 | x3; copy what is to be kept into profiles/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${TAG:-r6}; MODE=${MODE:-h2}; P=${TAG}_${MODE}
export TRIS_GEMM_MODE=$MODE
B="python bench.py --steps 2 --warmup 1 --headline-only"
# 1. kernel trace of the production configuration (autotuned, three streams, the step replayed from its graphs), the timed steps only.
#    Under the tracer neither way of issuing runs at speed: graph launches are serialised (40.6 ms/step with 7 ms of compute-stream gaps,
#    against 37.0 untraced), eager launches make the host the bottleneck (49 ms/step); the kernels and their durations are the same.
timeout 500 rocprofv3 --kernel-trace -d gpurun_out/${P}_trace -- python bench.py --steps 8 --warmup 3 --headline-only > gpurun_out/${P}_trace.log 2>&1 < /dev/null
DB=$(ls gpurun_out/${P}_trace/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB gpurun_out/${P}_kernel_stats.csv 30 6 > gpurun_out/${P}_kernel_stats_summary.txt < /dev/null; python tools/stream_gaps.py $DB 6 > gpurun_out/${P}_stream_gaps.txt < /dev/null; fi
rm -rf gpurun_out/${P}_trace
# 2. PMC passes (separate runs, static tile choice, eager launches: the profiler serialises kernels and reads counters per dispatch)
export TRIS_AUTOTUNE=0 TRIS_STEP_GRAPH=0
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${P}_fetch -o f -- $B > gpurun_out/${P}_fetch.log 2>&1 < /dev/null; echo fetch $?
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${P}_write -o w -- $B > gpurun_out/${P}_write.log 2>&1 < /dev/null; echo write $?
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${P}_sq -o s -- $B > gpurun_out/${P}_sq.log 2>&1 < /dev/null; echo sq $?
F=$(ls gpurun_out/${P}_fetch/*/*counter_collection.csv gpurun_out/${P}_fetch/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls gpurun_out/${P}_write/*/*counter_collection.csv gpurun_out/${P}_write/*counter_collection.csv 2>/dev/null | head -1)
S=$(ls gpurun_out/${P}_sq/*/*counter_collection.csv gpurun_out/${P}_sq/*counter_collection.csv 2>/dev/null | head -1)
echo "F=$F W=$W S=$S"
if [ -n "$F" ] && [ -n "$W" ]; then
  STEPS=$(python -c "
import csv,sys
n=sum(1 for r in csv.DictReader(open('$F')) if 'adamw_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE'); print(max(1,n//2))" < /dev/null)
  python tools/pmc_summary.py $F $W $STEPS gpurun_out/${P}_pmc_hbm_traffic.csv gpurun_out/${P}_pmc_hbm_traffic.json > gpurun_out/${P}_pmc_hbm.txt < /dev/null
fi
if [ -n "$S" ]; then python tools/pmc_sq_summary.py $S gpurun_out/${P}_pmc_mfma_util.csv gpurun_out/${P}_pmc_mfma_util.json > gpurun_out/${P}_pmc_sq.txt < /dev/null; fi
rm -rf gpurun_out/${P}_fetch gpurun_out/${P}_write gpurun_out/${P}_sq
tail -3 gpurun_out/${P}_pmc_hbm.txt gpurun_out/${P}_pmc_sq.txt 2>/dev/null | cut -c1-200
