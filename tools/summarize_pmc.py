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

# FETCH_SIZE is not bytes: on gfx950 it reads 0.500 of the bytes of a 16-byte-per-lane coalesced
# stream and 0.547 of a bit reader's (a dword per lane at a lane stride of ~36 bytes, the row
# kernels' payload reads); WRITE_SIZE reads 1.000 either way.  Both factors are MEASURED on known
# byte counts by tools/micro/hbm_calib under rocprofv3 (tools/calibrate_pmc.sh ->
# profiles/r05_calibration.json, read below; the defaults are that file's values).  Every kernel of
# the engine is one of the two patterns (matched on the name in front of the template arguments);
# anything else is reported raw with both corrections beside it.
_DEFAULT_FACTORS = {"stream16": {"fetch": 0.5000, "write": 1.0}, "rowlike": {"fetch": 0.5474, "write": 1.0}}


def _load_factors():
    here = os.path.dirname(os.path.abspath(__file__))
    import glob
    files = sorted(glob.glob(os.path.join(here, "..", "profiles", "r[0-9][0-9]_calibration.json")))   # the newest round's
    path = files[-1] if files else ""
    try:
        f = json.load(open(path))["factors"]
        return {k: {"fetch": float(f[k]["fetch"]), "write": float(f[k]["write"])} for k in ("stream16", "rowlike")}
    except (OSError, KeyError, TypeError, ValueError):
        return _DEFAULT_FACTORS


FACTORS = _load_factors()
PATTERN = {
    # 16 bytes per lane, coalesced
    "k_lowres_avg": "stream16", "k_pix_fwd": "stream16", "k_tok_hist": "stream16", "k_emit": "stream16",
    "k_emit_t": "stream16", "k_tok": "stream16", "k_emit_tok": "stream16", "k_front": "stream16", "k_emit_m": "stream16", "k_lres_summary": "stream16", "k_place_fres": "stream16",
    "k_tile_inv": "stream16", "k_tile_fwd": "stream16", "k_cal_read": "stream16", "k_cal_copy": "stream16",
    # bit readers: a dword per lane along the lane's own sub-sequence of the payload
    "k_dec_row_fused": "rowlike", "k_row_count": "rowlike", "k_row_count_w": "rowlike", "k_row_count_q": "rowlike",
    "k_row_window": "rowlike", "k_dec_huff": "rowlike", "k_lres_spec": "rowlike", "k_lres_fix": "rowlike",
    "k_lres_write": "rowlike", "k_cal_rowlike": "rowlike",
}


def pattern_of(key):
    return PATTERN.get(key.split("<")[0].split("[")[0].strip().lstrip("("))


def is_streaming(key):
    return pattern_of(key) == "stream16"

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
    # FETCH_SIZE -> bytes by the measured factor of the kernel's access pattern (see FACTORS).
    if "FETCH_SIZE" in row:
        raw = row["FETCH_SIZE"] * 1024 / 1e6
        pat = pattern_of(k)
        row["hbm_read_MB_raw"] = raw
        row["hbm_read_MB_x2"] = 2 * raw
        row["hbm_read_MB_corrected"] = raw / FACTORS[pat]["fetch"] if pat else raw
        row["fetch_pattern"] = {"stream16": 16, "rowlike": 4}.get(pat, 0)   # bytes per lane of the calibrated pattern (0: uncalibrated, raw)
    if "WRITE_SIZE" in row:
        row["hbm_write_MB"] = row["WRITE_SIZE"] * 1024 / 1e6
    out[k] = row
    print(k)
    for c in sorted(row):
        print("   %-28s %16.1f" % (c, row[c]))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
