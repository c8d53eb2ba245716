"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files of a probe binary (several passes, one file each).
usage: probe_pmc_summary.py out.txt pass1.csv [pass2.csv ...]   -- SQ_* wave-level counters are printed relative to SQ_WAVE_CYCLES"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", r["Kernel_Name"])
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for n, cs in acc.items():
    if "to_planes" in n or "ref_rows" in n:
        continue
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    lines.append(f"{n}   ({len(next(iter(cs.values())))} dispatches)")
    for k in sorted(m):
        rel = f"  = {m[k] / wc:6.3f} of SQ_WAVE_CYCLES" if wc and k.startswith("SQ_") and k not in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA") else ""
        lines.append(f"    {k:32s} {m[k]:16.0f}{rel}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        lines.append(f"    matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:80]))
