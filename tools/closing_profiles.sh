# final measurement set of a round (TAG=r3z bash tools/closing_profiles.sh) (run on the GPU box through gpurun); outputs under gpurun_out/${TAG}_*
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${TAG:-r3z}
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline"
# 1. kernel trace of the production configuration (autotuned, three streams)
timeout 500 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_trace -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-pipeline > gpurun_out/${TAG}_trace.log 2>&1 < /dev/null
DB=$(ls gpurun_out/${TAG}_trace/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB gpurun_out/${TAG}_kernel_stats_x3_autotuned.csv 20 6 > gpurun_out/${TAG}_kernel_stats.txt < /dev/null; python tools/stream_gaps.py $DB 6 > gpurun_out/${TAG}_stream_gaps.txt < /dev/null; fi
rm -rf gpurun_out/${TAG}_trace
# 2. PMC passes (separate runs, static tile choice, kernels serialised by the profiler)
export TRIS_AUTOTUNE=0
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -o f -- $B > gpurun_out/${TAG}_fetch.log 2>&1 < /dev/null; echo fetch $?
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -o w -- $B > gpurun_out/${TAG}_write.log 2>&1 < /dev/null; echo write $?
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_sq -o s -- $B > gpurun_out/${TAG}_sq.log 2>&1 < /dev/null; echo sq $?
F=$(ls gpurun_out/${TAG}_fetch/*/*counter_collection.csv gpurun_out/${TAG}_fetch/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls gpurun_out/${TAG}_write/*/*counter_collection.csv gpurun_out/${TAG}_write/*counter_collection.csv 2>/dev/null | head -1)
S=$(ls gpurun_out/${TAG}_sq/*/*counter_collection.csv gpurun_out/${TAG}_sq/*counter_collection.csv 2>/dev/null | head -1)
echo "F=$F W=$W S=$S"
if [ -n "$F" ] && [ -n "$W" ]; then
  STEPS=$(python -c "
import csv,sys
n=sum(1 for r in csv.DictReader(open('$F')) if 'adamw_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE'); print(max(1,n//2))" < /dev/null)
  python tools/pmc_summary.py $F $W $STEPS gpurun_out/${TAG}_pmc_hbm_traffic.csv gpurun_out/${TAG}_pmc_hbm_traffic.json > gpurun_out/${TAG}_pmc_hbm.txt < /dev/null
fi
if [ -n "$S" ]; then python tools/pmc_sq_summary.py $S gpurun_out/${TAG}_pmc_mfma_util.csv gpurun_out/${TAG}_pmc_mfma_util.json > gpurun_out/${TAG}_pmc_sq.txt < /dev/null; fi
rm -rf gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sq
tail -3 gpurun_out/${TAG}_pmc_hbm.txt gpurun_out/${TAG}_pmc_sq.txt 2>/dev/null | cut -c1-200
