cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 120 tools/probes/h2_phase_probe > gpurun_out/r6_phase_probe_a.txt 2>&1; echo "probe rc $?" ) 
( timeout 120 tools/probes/mfma_lds_overlap > gpurun_out/r6_mfma_lds_overlap.txt 2>&1; echo "overlap rc $?" )
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "fused_splitk" > gpurun_out/r6_tests_a.txt 2>&1; echo "t1 rc $?"
timeout 600 python -m pytest tests/test_gpu_planes.py -q -x -k "two_forwards or bottleneck_stack" >> gpurun_out/r6_tests_a.txt 2>&1; echo "t2 rc $?"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipeline > gpurun_out/r6_bench_a.json 2> gpurun_out/r6_bench_a.err; echo "bench rc $?"
TRIS_FUSE_SPLITK_PY=0 timeout 300 python bench.py --steps 10 --warmup 3 --headline-only > gpurun_out/r6_bench_a_nofuse.json 2>> gpurun_out/r6_bench_a.err; echo "bench nofuse rc $?"
timeout 300 python bench.py --steps 10 --warmup 3 --headline-only > gpurun_out/r6_bench_a_fuse.json 2>> gpurun_out/r6_bench_a.err; echo "bench fuse rc $?"
timeout 400 rocprofv3 --kernel-trace -d gpurun_out/r6a_trace -- python bench.py --steps 8 --warmup 3 --headline-only > gpurun_out/r6a_trace.log 2>&1 < /dev/null
DB=$(ls gpurun_out/r6a_trace/*/*.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB gpurun_out/r6a_kernel_stats.csv 40 6 > gpurun_out/r6a_kernel_stats_summary.txt < /dev/null; python tools/stream_gaps.py $DB 6 > gpurun_out/r6a_stream_gaps.txt < /dev/null; python tools/step_sequence.py $DB 2 > gpurun_out/r6a_step_sequence.txt < /dev/null; fi
rm -rf gpurun_out/r6a_trace
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "headline_batch_48" > gpurun_out/r6_tests_b.txt 2>&1; echo "t3 rc $?"
tail -3 gpurun_out/r6_tests_a.txt gpurun_out/r6_tests_b.txt; cat gpurun_out/r6_bench_a_nofuse.json gpurun_out/r6_bench_a_fuse.json; head -c 600 gpurun_out/r6_bench_a.json
