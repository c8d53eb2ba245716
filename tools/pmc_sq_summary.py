"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass (counter_collection.csv).  SQ_VALU_MFMA_BUSY_CYCLES is the chip-wide sum of matrix-pipe busy cycles (32 per
v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so
    MFMA utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).
usage: pmc_sq_summary.py s_counter_collection.csv out.csv out.json"""
import collections, csv, json, sys
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in rows:
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        calls[r["Kernel_Name"]] += 1
steps = max(1, calls.get(next((k for k in calls if "adamw_kernel" in k), ""), 2) // 2)
def util(d):
    return d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if d["GRBM_GUI_ACTIVE"] else 0.0
out = sorted(((k, calls[k], d) for k, d in agg.items()), key=lambda t: -t[2]["GRBM_GUI_ACTIVE"])
with open(sys.argv[2], "w") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel", "LaunchesPerStep", "GpuActiveCyclesPerStep(per XCD)", "MfmaBusyCyclesPerStep", "MfmaUtilisation"])
    for k, c, d in out:
        w.writerow([k, round(c / steps, 2), int(d["GRBM_GUI_ACTIVE"] / 8 / steps), int(d["SQ_VALU_MFMA_BUSY_CYCLES"] / steps),
                    round(util(d), 4)])
fam = collections.defaultdict(lambda: collections.defaultdict(float))
for k, c, d in out:
    f = "gemm_family" if ("gemm_" in k or "wgrad3x3_direct" in k) else "xattn" if "xattn" in k else "mha_mfma" if "mha_mfma" in k else "other"
    for n, v in d.items():
        fam[f][n] += v
summ = {"steps_profiled": steps,
        "families": {f: {"mfma_utilisation": round(util(d), 4), "mfma_busy_cycles_per_step": int(d["SQ_VALU_MFMA_BUSY_CYCLES"] / steps),
                         "gpu_active_cycles_per_step_per_xcd": int(d["GRBM_GUI_ACTIVE"] / 8 / steps)} for f, d in fam.items()},
        "method": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE on "
                  "`TRIS_AUTOTUNE=0 python bench.py --steps 2 --warmup 1`; utilisation = MFMA_BUSY / (GUI_ACTIVE/8 * 1024)"}
json.dump(summ, open(sys.argv[3], "w"), indent=1)
print(json.dumps(summ, indent=1))
for k, c, d in out[:14]:
    print(f"{util(d)*100:6.1f}%  {c/steps:7.1f}/step  {k[:110]}")
