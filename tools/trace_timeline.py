#!/usr/bin/env python3
"""Timeline of the LAST repetition in a rocprofv3 --kernel-trace CSV: every kernel's start
(relative to the repetition's first kernel), duration and queue.
  trace_timeline.py <dir or *_kernel_trace.csv> <name of the repetition's first kernel>"""
import csv, glob, os, sys
src = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_dec_zero"
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].split("(")[0].split("<")[0].endswith(first)]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
end = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    end = max(end, e)
    name = r["Kernel_Name"].split("(")[0].replace("himg_dev::", "").replace("void ", "")
    print("%9.1f us  +%8.1f us  q%-3s grid %-9s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), name[:60]))
print("repetition: %.1f us" % (end / 1e3))
