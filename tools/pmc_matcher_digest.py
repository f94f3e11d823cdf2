"""Digest the matcher's --pmc passes of tools/profile_round.sh <tag> into profiles/<tag>_matcher_pmc.json (read back by
bench.py into roofline.matcher.config4_2048x2048.mfma_counters) and profiles/<tag>_matcher_counters.txt.
usage: python tools/pmc_matcher_digest.py <tag>"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
kname = "k_ov_match"
for d in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmcm_{tag}_*")):
    if not os.path.isdir(d):
        continue
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        if "k_ov_match" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
            kname = "k_ov_match_f4" if "k_ov_match_f4" in r["Kernel_Name"] else ("k_ov_match_sp" if "k_ov_match_sp" in r["Kernel_Name"] else "k_ov_match")
avg = {k: s / n for k, (s, n) in agg.items()}
fp4 = kname == "k_ov_match_f4"
out = {"kernel": kname, "operands": "fp4 E2M1 (v_mfma_scale_f32_16x16x128_f8f6f4)" if fp4 else "i8 (v_mfma_i32_16x16x64_i8)", "workload": "64 pairs x (2048 x 2048) 512-bit descriptors, per launch", "counters_per_launch": avg,
       "source": f"rocprofv3 --pmc (two separate passes) -- python3 tools/matcher_only.py; tools/profile_round.sh {tag}"}
if "SQ_INSTS_MFMA" in avg and "SQ_INSTS_VALU" in avg:
    out["valu_per_mfma"] = (avg["SQ_INSTS_VALU"] - avg["SQ_INSTS_MFMA"]) / avg["SQ_INSTS_MFMA"] if avg["SQ_INSTS_VALU"] > avg["SQ_INSTS_MFMA"] else avg["SQ_INSTS_VALU"] / avg["SQ_INSTS_MFMA"]
    out["valu_per_mfma_note"] = "SQ_INSTS_VALU includes the MFMA instructions on this chip when it exceeds them (subtracted); else the plain ratio"
    # one v_mfma_i32_16x16x64_i8 = 16384 MACs, one v_mfma_scale_f32_16x16x128_f8f6f4 = 32768
    out["mfma_wave_instr_expected"] = 64 * 2048 * 2048 * 512 / (16 * 16 * (128 if fp4 else 64))
if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_INSTS_MFMA" in avg:
    out["mfma_busy_cycles_per_mfma"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / avg["SQ_INSTS_MFMA"]
if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg:
    # the matrix pipes of the chip's 1024 SIMDs against the cycles the launch kept the GPU active
    # GRBM_GUI_ACTIVE comes summed over the 8 XCDs (8 x the launch's cycles); an XCD has 32 CUs x 4 SIMDs
    out["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] * 128.0)
    out["launch_cycles_per_xcd"] = avg["GRBM_GUI_ACTIVE"] / 8.0
    out["mfma_busy_frac_note"] = ("matrix-pipe busy cycles of all SIMDs / (cycles the launch kept the chip active x 1024 SIMDs): "
                                  "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE [summed over 8 XCDs] x 128 SIMDs per XCD); against the "
                                  "clock the chip actually held, where `frac` prices the same launch against the 2.4 GHz peak")
try:
    m = json.loads([l for l in open(os.path.join(ROOT, "gpurun_out", f"matcher_{tag}.json")) if l.startswith("{")][-1])
    out["unprofiled_run"] = {k: m[k] for k in ("avg_launch_ms", "achieved", "frac", "frac_of_i8_peak", "frac_of_fp4_peak", "form") if k in m}
except Exception:
    pass
try:
    import subprocess, time
    out["_meta"] = {"collected": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                    "git_head": subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip(),
                    "dirty": bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "uwimageproc_amd", "bench.py", "include"], text=True).strip())}
except Exception:
    out["_meta"] = {}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_matcher_pmc.json"), "w"), indent=1, sort_keys=True)
with open(os.path.join(ROOT, "profiles", f"{tag}_matcher_counters.txt"), "w") as f:
    for k in sorted(avg):
        f.write(f"{k:28s} {avg[k]:18.1f}  (n={agg[k][1]})\n")
print(json.dumps(out, indent=1))
