#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_passes.sh for the dominant kernel (default conv_split_kernel<9,...>;
third argument = another kernel-name substring, e.g. "conv_mfma_kernel<9" for CDDPM_CONV=f32 runs).

Applies the gfx950 corrections of MI355X_MICROARCH.md: FETCH_SIZE (KB) under-reports wide (16 B/lane) coalesced
reads by exactly 2x -> doubled; WRITE_SIZE (KB) is exact for 16-B-per-lane stores. Writes
profiles/<round>_conv3x3_hbm_traffic.json (read by bench.py for roofline.traffic, labelled with its source) and prints the utilisation figures.
usage: python tools/pmc_summary.py gpurun_out profiles/r01_pmc_summary.json [kernel-substring]"""
import collections
import csv
import json
import os
import sys

root, out = sys.argv[1], sys.argv[2]
KEY = sys.argv[3] if len(sys.argv) > 3 else "conv_split_kernel<9"
SUF = sys.argv[4] if len(sys.argv) > 4 else ""         # directory suffix of tools/pmc_passes.sh <suffix>


def load(tag):
    path = os.path.join(root, f"pmc{SUF}_{tag}", "pmc_counter_collection.csv")
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if KEY in r["Kernel_Name"]:
            d = per[r["Dispatch_Id"]]
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return list(per.values())


res = {}
a = load("SQ_VALU_MFMA_BUSY_CYCLES")
n = len(a)
tot = lambda rows, k: sum(r.get(k, 0.0) for r in rows)
ns = tot(a, "ns")
gui = tot(a, "GRBM_GUI_ACTIVE") / 8.0                 # summed over 8 XCDs
res["launches"] = n
res["avg_launch_us"] = ns / n / 1e3
res["effective_clock_GHz"] = gui / ns
res["kernel"] = KEY
res["mfma_pipe_busy_frac"] = tot(a, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * gui)   # 256 CUs x 4 SIMDs
res["wave_cycles_split"] = {k: tot(a, k) / tot(a, "SQ_WAVE_CYCLES") for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
b = load("SQ_LDS_BANK_CONFLICT")
res["lds_bank_conflict_frac"] = tot(b, "SQ_LDS_BANK_CONFLICT") / max(1.0, tot(b, "SQ_LDS_IDX_ACTIVE"))
res["valu_per_mfma"] = (tot(b, "SQ_INSTS_VALU") - tot(b, "SQ_INSTS_MFMA")) / max(1.0, tot(b, "SQ_INSTS_MFMA"))
f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
fetch = 2.0 * tot(f, "FETCH_SIZE") * 1024.0 / len(f)   # x2: gfx950 wide-read correction
write = tot(w, "WRITE_SIZE") * 1024.0 / len(w)
res["hbm_fetch_bytes_per_launch"] = fetch
res["hbm_write_bytes_per_launch"] = write
res["bytes_per_launch"] = fetch + write
json.dump(res, open(out, "w"), indent=1)
if KEY.startswith("conv_split_kernel<9") and not SUF:       # only the headline kernel's default run feeds bench.py's roofline.traffic
  json.dump({"bytes_per_launch": fetch + write, "fetch": fetch, "write": write,
             "note": "rocprofv3 FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, average over the conv3x3 launches of 50 reverse steps, B=64 128x128"},
            open(os.path.join(os.path.dirname(out), os.path.basename(out).split("_")[0] + "_conv3x3_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
