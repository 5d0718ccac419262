"""Condense a profiles/tools/collect_sq.sh run: per kernel (averaged over its launches) the SQ / GRBM counters, the derived
MFMA-busy fraction, wait fractions and the effective clock.  usage: summarise_sq.py gpurun_out/sq_<tag>
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES
count cycles (MI355X_MICROARCH.md, cycle constants); GRBM_GUI_ACTIVE is the sum over the 8 XCDs."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
res = {}
for wl in sorted(os.listdir(src)):
    wdir = os.path.join(src, wl)
    if not os.path.isdir(wdir):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(wdir, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in r:
                d = dur[r["Kernel_Name"]]
                d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                d[1] += 1
    rows = []
    for k, cs in acc.items():
        c = {n: v[0] / v[1] for n, v in cs.items() if v[1]}
        n = max(v[1] for v in cs.values())
        row = {"kernel": k, "launches_seen": n, "counters": c}
        if dur[k][1]:
            ns = dur[k][0] / dur[k][1]
            row["avg_ns_in_grbm_pass"] = ns
            if "GRBM_GUI_ACTIVE" in c and ns > 0:
                row["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / ns
        if c.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # SQ_BUSY_CYCLES: per-SE busy cycles summed; MFMA busy is summed over SIMDs -> normalise by GRBM cycles when present
            pass
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for nme in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if nme in c:
                    row[nme.lower() + "_frac"] = c[nme] / wc
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                # waves are one per SIMD in the contraction kernels: MFMA-busy cycles / (4 x wave quad-cycles) = pipe utilisation
                row["mfma_busy_per_wave_cycle"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_conflict_ratio"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        rows.append(row)
    rows.sort(key=lambda r: -r["counters"].get("SQ_WAVE_CYCLES", r["counters"].get("GRBM_GUI_ACTIVE", 0.0)) * r["launches_seen"])
    res[wl] = rows[:24]
json.dump(res, open(os.path.join(src, "sq_summary.json"), "w"), indent=1)
for wl, rows in res.items():
    print("==", wl)
    for r in rows[:12]:
        print("%-78s n=%-3d clk=%s mfma/wave=%s wait_any=%s wait_inst=%s active=%s" % (
            r["kernel"][:78], r["launches_seen"], "%.2f" % r["clock_ghz"] if "clock_ghz" in r else "-",
            "%.3f" % r["mfma_busy_per_wave_cycle"] if "mfma_busy_per_wave_cycle" in r else "-",
            "%.3f" % r["sq_wait_any_frac"] if "sq_wait_any_frac" in r else "-",
            "%.3f" % r["sq_wait_inst_any_frac"] if "sq_wait_inst_any_frac" in r else "-",
            "%.3f" % r["sq_active_inst_any_frac"] if "sq_active_inst_any_frac" in r else "-"))
