#!/usr/bin/env python3
"""Dev tool: condense gpurun_out/prof_<tag>/ (rocprofv3 CSVs from tools/profile.sh) into the
tracked summaries under profiles/:  <tag>_kernel_stats.csv, <tag>_pmc.csv, <tag>_summary.md and
latest_pmc.json (read by bench.py for roofline.traffic)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def find(pattern):
    return sorted(glob.glob(os.path.join(src, pattern), recursive=True))


# ---- kernel stats (rocprofv3 --kernel-trace --stats) ----
stats = find("trace/**/*kernel_stats.csv")
rows = []
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write(open(stats[0]).read())

# ---- PMC passes ----
agg = defaultdict(lambda: defaultdict(list))
for path in find("pmc*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pmc_rows = []
for k, cs in agg.items():
    for c, v in sorted(cs.items()):
        pmc_rows.append((k, c, len(v), sum(v) / len(v)))
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for r in pmc_rows:
        f.write('"%s",%s,%d,%.6g\n' % r)

main = next((k for k in agg if "fwd_stream" in k), None)
out = {"tag": tag}
md = [f"# rocprofv3 summary `{tag}` (python bench.py --steps 30 --warmup 5, hc 10k x 128, 1x MI355X)", ""]
if rows:
    md += ["## kernel-trace --stats", "", "| kernel | calls | avg ns | total ns | % |", "|---|---|---|---|---|"]
    for r in rows[:8]:
        md.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {r['AverageNs']} | {r['TotalDurationNs']} | {r['Percentage']} |")
    for r in rows:
        if "fwd_stream" in r["Name"]:
            out["main_kernel_avg_ms"] = float(r["AverageNs"]) * 1e-6
if main:
    c = {k: sum(v) / len(v) for k, v in agg[main].items()}
    fetch_kb, write_kb = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    if fetch_kb is not None and write_kb is not None:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half
        # of the bytes of wide coalesced reads -> doubled (upper bound for this kernel's narrow loads).
        out["fetch_size_kb"] = fetch_kb
        out["write_size_kb"] = write_kb
        out["hbm_bytes_per_launch"] = (2 * fetch_kb + write_kb) * 1024
    md += ["", f"## PMC, `{main[:80]}` (mean per dispatch)", "", "| counter | value |", "|---|---|"]
    for k, v in sorted(c.items()):
        md.append(f"| {k} | {v:.6g} |")
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
        out["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
        out["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)
    if all(k in c for k in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")):
        w = c["SQ_WAVE_CYCLES"]
        md += ["", f"wave-cycle split: issuing {c['SQ_ACTIVE_INST_ANY']/w:.1%}, issue-stalled {c['SQ_WAIT_INST_ANY']/w:.1%}, "
                   f"waiting (waitcnt) {c['SQ_WAIT_ANY']/w:.1%}"]
# ---- per dispatch: duration and GRBM_GUI_ACTIVE of the main kernel (why dispatches of the same work differ) ----
if main:
    per = defaultdict(dict)
    for path in find("pmc*/**/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Kernel_Name"] != main or r["Counter_Name"] not in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
                continue
            key = (path, r["Dispatch_Id"])
            per[key][r["Counter_Name"]] = float(r["Counter_Value"])
            per[key]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    rows_d = [v for v in per.values() if "GRBM_GUI_ACTIVE" in v]
    if rows_d:
        md += ["", "## per dispatch (the PMC pass that holds GRBM_GUI_ACTIVE; the counter sums the 8 XCDs)", "",
               "| dispatch | ms | GRBM_GUI_ACTIVE | clock GHz = GUI_ACTIVE / 8 / time | SQ_WAVE_CYCLES |", "|---|---|---|---|---|"]
        for i, v in enumerate(rows_d):
            md.append(f"| {i} | {v['ms']:.3f} | {v['GRBM_GUI_ACTIVE']:.5g} | {v['GRBM_GUI_ACTIVE'] / 8 / (v['ms'] * 1e-3) * 1e-9:.3f} | {v.get('SQ_WAVE_CYCLES', 0):.5g} |")
        ms = [v["ms"] for v in rows_d]
        cyc = [v["GRBM_GUI_ACTIVE"] for v in rows_d]
        out["dispatch_ms_min_max"] = [min(ms), max(ms)]
        out["gui_active_min_max"] = [min(cyc), max(cyc)]
        out["effective_clock_ghz"] = sum(cyc) / 8 / (sum(ms) * 1e-3) * 1e-9
        md += ["", f"time spread {min(ms):.3f}..{max(ms):.3f} ms ({max(ms) / min(ms) - 1:.1%}), busy-cycle spread "
                   f"{max(cyc) / min(cyc) - 1:.1%}: what is left of the time spread after the cycle spread is clock (power management), "
                   "the cycle spread itself is dispatch order / tail."]
if main and "SQ_INSTS_VALU" in c:
    out["valu_per_cell_note"] = "SQ_INSTS_VALU x 64 / cells_per_gpu of the bench line"
json.dump(out, open(os.path.join(dst, "latest_pmc.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(md) + "\n")
bj = os.path.join(src, "bench_under_trace.json")
if os.path.exists(bj):
    open(os.path.join(dst, f"{tag}_bench_under_trace.json"), "w").write(open(bj).read())
print("\n".join(md))
print(json.dumps(out))
