cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_tt -o t -- python bench.py --steps 4 --warmup 2 --headline-only > gpurun_out/_tt.log 2>&1 < /dev/null
F=$(ls gpurun_out/_tt/*/*kernel_trace.csv gpurun_out/_tt/*kernel_trace.csv 2>/dev/null | head -1)
python - "$F" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "launches; columns:", list(rows[0].keys())[:14])
def nm(r): return re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", r["Kernel_Name"])[:44]
last = rows[-len(rows)//6:]   # roughly the last step
sel = [r for r in last if any(k in r["Kernel_Name"] for k in ("splitk_reduce", "finalize", "slab_reduce", "col_partial", "layernorm_bwd"))]
sel.sort(key=lambda r: -(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for r in sel[:25]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{d:8.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8s} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size','?')):>5s}  queue {r.get('Queue_Id','?'):>3s}  {nm(r)}")
PY
rm -rf gpurun_out/_tt gpurun_out/_tt.log
