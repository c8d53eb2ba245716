# PMC passes for the cross-attention forward alone (GPU box): HBM-side traffic and L2 hit rate per launch of each form
# usage: bash tools/xattn_pmc.sh [tag]   -> gpurun_out/<tag>_xattn_pmc.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r4}
OUT=gpurun_out/${TAG}_xattn_pmc.txt; : > $OUT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=gpurun_out/_xp_$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -o x -- python tools/xattn_check.py > $D.log 2>&1 < /dev/null
  F=$(ls $D/*/*counter_collection.csv $D/*counter_collection.csv 2>/dev/null | head -1)
  python - "$F" >> $OUT <<'PY'
import collections, csv, sys
tot, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "xattn" not in r["Kernel_Name"]:
        continue
    import re
    k = (re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", r["Kernel_Name"])[:62], r["Counter_Name"]); tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for (k, c), v in sorted(tot.items()):
    print(f"{c:14s} {k:62s} launches {cnt[(k, c)]:4d}  per launch {v / cnt[(k, c)]:14.1f}")
PY
  rm -rf $D $D.log
done
# kernel durations of the same script (idle device): rocprofv3 --kernel-trace --stats
D=gpurun_out/_xp_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o x -- python tools/xattn_check.py > $D.log 2>&1 < /dev/null
F=$(ls $D/*/*kernel_stats.csv $D/*kernel_stats.csv 2>/dev/null | head -1)
echo "kernel durations (rocprofv3 --kernel-trace --stats, ns): name, calls, average, min, max" >> $OUT
python - "$F" >> $OUT <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "xattn" in r["Name"]:
        n = re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", r["Name"])[:50]
        print(f"  {n:52s} {r['Calls']:>5s} {float(r['AverageNs']):10.0f} {float(r['MinNs']):10.0f} {float(r['MaxNs']):10.0f}")
PY
rm -rf $D $D.log
cat $OUT
