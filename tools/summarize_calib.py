#!/usr/bin/env python3
"""Counter calibration from tools/calibrate_pmc.sh: FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3
reports them) of tools/micro/hbm_calib's kernels against the bytes those kernels are known to
move.  Output: one JSON object with, per access pattern, counter bytes / true bytes -- the
factor tools/summarize_pmc.py divides by instead of a hand-kept list of "streaming" kernels:
  stream16   16 bytes per lane, coalesced (k_cal_read / k_cal_write / k_cal_copy)
  rowlike    the row kernel's pattern: a dword per lane at ~36-byte lane stride, 2 x 16-byte
             stores per lane and pixel row (k_cal_rowlike)
"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
run = json.load(open(os.path.join(root, "hbm_calib.json")))   # the unprofiled run: rates
GIB = 2.0   # the profiled runs use 2 GiB buffers (calibrate_pmc.sh)
nbytes = GIB * 1024 ** 3
row_bytes = 8 * 4096 * 4
rows = int(nbytes // row_bytes)
true = {
    "k_cal_read": {"rd": nbytes, "wr": 0.0},
    "k_cal_write": {"rd": 0.0, "wr": nbytes},
    "k_cal_copy": {"rd": nbytes, "wr": nbytes},
    "k_cal_rowlike": {"rd": rows * 33600.0, "wr": rows * float(row_bytes)},
}
meas = {k: {"FETCH_SIZE": [], "WRITE_SIZE": []} for k in true}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(os.path.join(root, c, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].split()[-1]
            if name in meas and r["Counter_Name"] == c:
                meas[name][c].append(float(r["Counter_Value"]) * 1024.0)
out = {"buffer_GiB_profiled": GIB, "rates_unprofiled": run, "kernels": {}}
for k, t in true.items():
    f = sorted(meas[k]["FETCH_SIZE"])
    w = sorted(meas[k]["WRITE_SIZE"])
    # the median dispatch (the program sweeps grid sizes: every dispatch moves the same bytes)
    fm = f[len(f) // 2] if f else None
    wm = w[len(w) // 2] if w else None
    out["kernels"][k] = {
        "true_read_bytes": t["rd"], "true_write_bytes": t["wr"],
        "FETCH_SIZE_bytes": fm, "WRITE_SIZE_bytes": wm,
        "fetch_over_true": (fm / t["rd"]) if fm is not None and t["rd"] else None,
        "write_over_true": (wm / t["wr"]) if wm is not None and t["wr"] else None,
        "dispatches": len(f),
    }
ks = out["kernels"]
def avg(vals):
    vals = [v for v in vals if v]
    return sum(vals) / len(vals) if vals else None
out["factors"] = {
    "stream16": {"fetch": avg([ks["k_cal_read"]["fetch_over_true"], ks["k_cal_copy"]["fetch_over_true"]]),
                 "write": avg([ks["k_cal_write"]["write_over_true"], ks["k_cal_copy"]["write_over_true"]])},
    "rowlike": {"fetch": ks["k_cal_rowlike"]["fetch_over_true"], "write": ks["k_cal_rowlike"]["write_over_true"]},
}
print(json.dumps(out, indent=1))
