#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel (mean per dispatch)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def kname(full):
    """'void himg_dev::k_x<8, true>(himg_dev::Geom, ...)' -> 'k_x<8, true>'"""
    n = full.replace("himg_dev::", "")
    if n.startswith("void "):
        n = n[5:]
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i]
    return n

# Kernels that read their input as 16-byte-per-lane coalesced streams (matched on the
# name in front of the template arguments; k_emit_t<8> / k_emit_t<1> are "k_emit_t").
STREAMING = {"k_lowres_avg", "k_pix_fwd", "k_tok_hist", "k_emit", "k_emit_t", "k_emit_m", "k_lres_summary",
             "k_place_fres", "k_tile_inv"}


def is_streaming(key):
    return key.split("<")[0].split("[")[0].strip() in STREAMING

# A kernel launched with several grid sizes per step (k_tok_hist: the LRES spans on the
# side stream, then the FRES rows) is reported per grid: the largest under the kernel's
# name, the others as "name[grid N]" -- a mean over both would halve every per-launch
# figure of the launch that matters.
def grid_of(r):
    if "Grid_Size" in r:
        return int(r["Grid_Size"])
    return int(r.get("Grid_Size_X", 1)) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))


rows_c, rows_t = [], []
for path in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    rows_c += [(kname(r["Kernel_Name"]), grid_of(r), r) for r in csv.DictReader(open(path))]
for path in glob.glob(os.path.join(root, "*", "*", "*kernel_trace.csv")):
    rows_t += [(kname(r["Kernel_Name"]), grid_of(r), r) for r in csv.DictReader(open(path))]
gmax = collections.defaultdict(int)
for k, g, _ in rows_c + rows_t:
    gmax[k] = max(gmax[k], g)


def key(k, g):
    return k if g == gmax[k] else "%s[grid %d]" % (k, g)


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for k, g, r in rows_c:
    a = acc[key(k, g)][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
for k, g, r in rows_t:
    d = dur[key(k, g)]
    d[0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    d[1] += 1
out = {}
for k in sorted(acc, key=lambda k: -dur[k][0]):
    if not k.startswith("k_"):
        continue
    row = {c: v[0] / max(v[1], 1) for c, v in acc[k].items()}
    row["dur_us"] = dur[k][0] / max(dur[k][1], 1)
    # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads HALF the bytes of wide coalesced
    # streaming reads (16 B per lane); other access widths are uncalibrated.  So the x2
    # is applied only to the kernels whose reads are 16-byte-per-lane streams
    # (STREAMING below); for the others the raw figure is reported, with the x2 value
    # beside it as an upper bound.
    if "FETCH_SIZE" in row:
        raw = row["FETCH_SIZE"] * 1024 / 1e6
        row["hbm_read_MB_raw"] = raw
        row["hbm_read_MB_x2"] = 2 * raw
        row["hbm_read_MB_corrected"] = 2 * raw if is_streaming(k) else raw
    if "WRITE_SIZE" in row:
        row["hbm_write_MB"] = row["WRITE_SIZE"] * 1024 / 1e6
    out[k] = row
    print(k)
    for c in sorted(row):
        print("   %-28s %16.1f" % (c, row[c]))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
