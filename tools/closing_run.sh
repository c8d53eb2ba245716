# closing run of a round (on the GPU box: gpurun -- bash tools/closing_run.sh): full GPU test suite, smoke, bench (with CPU baseline and every side leg; autotuner log per arithmetic), the cross-attention phase table, then the profile sets of tools/closing_profiles.sh for h2 and x3; outputs under gpurun_out/, copied to profiles/ by hand
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export TAG=${TAG:-r6}
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_gputests.log 2>&1 < /dev/null; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_gputests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1 < /dev/null; tail -1 gpurun_out/${TAG}_smoke.log
TRIS_TUNE_LOG=gpurun_out/${TAG}_autotune_log_all.txt timeout 1200 python bench.py > gpurun_out/${TAG}_bench.log 2>&1 < /dev/null; grep "^{" gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json; cut -c1-300 gpurun_out/${TAG}_bench.json
grep -E "mode=3|mode=4" gpurun_out/${TAG}_autotune_log_all.txt > gpurun_out/${TAG}_autotune_log_h2.txt; grep "mode=1" gpurun_out/${TAG}_autotune_log_all.txt > gpurun_out/${TAG}_autotune_log_x3.txt; wc -l gpurun_out/${TAG}_autotune_log_*.txt | tail -3
timeout 300 python tools/xattn_px_trace.py > gpurun_out/${TAG}_xattn_phase_trace.txt 2>&1 < /dev/null; tail -18 gpurun_out/${TAG}_xattn_phase_trace.txt | cut -c1-200
timeout 300 python tools/xattn_check.py > gpurun_out/${TAG}_xattn_forms.txt 2>&1 < /dev/null; tail -3 gpurun_out/${TAG}_xattn_forms.txt | cut -c1-200
bash tools/xattn_pmc.sh ${TAG} > /dev/null 2>&1 < /dev/null; cut -c1-160 gpurun_out/${TAG}_xattn_pmc.txt
timeout 300 python tools/step_graph_marks.py > gpurun_out/${TAG}_step_graph_marks.txt 2>&1 < /dev/null; tail -2 gpurun_out/${TAG}_step_graph_marks.txt
timeout 300 python tools/wait_probe.py 5 > gpurun_out/${TAG}_wait_probe.txt 2>&1 < /dev/null; grep -v 'stream2  wait' gpurun_out/${TAG}_wait_probe.txt | tail -8
# the ViT-B/16 trunk (BASELINE configs[4]): kernel trace of the timed steps
timeout 700 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_vit_trace -- python bench.py --backbone clip-ViT-B/16 --steps 6 --warmup 2 --headline-only > gpurun_out/${TAG}_vit_b16_trace.log 2>&1 < /dev/null
DB=$(ls gpurun_out/${TAG}_vit_trace/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB gpurun_out/${TAG}_vit_b16_kernel_stats.csv 30 4 > gpurun_out/${TAG}_vit_b16_kernel_stats_summary.txt < /dev/null; python tools/stream_gaps.py $DB 4 > gpurun_out/${TAG}_vit_b16_stream_gaps.txt < /dev/null; fi
rm -rf gpurun_out/${TAG}_vit_trace; grep "^{" gpurun_out/${TAG}_vit_b16_trace.log | cut -c1-200
MODE=h2 bash tools/closing_profiles.sh
MODE=x3 bash tools/closing_profiles.sh
