#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile_pmc.sh summary (bytes per frame)."""
import json
import sys

summary, frames_per_launch, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
d = json.load(open(summary))
t = {k: round((v.get("hbm_read_MB_corrected", 0) + v.get("hbm_write_MB", 0)) * 1e6 / frames_per_launch)
     for k, v in d.items()}
json.dump({"note": "HBM bytes per 4096x4096 RGBA q50 randtile frame per kernel launch: rocprofv3 --pmc "
                   "FETCH_SIZE (x2, gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE, separate passes, "
                   "%d frames per launch, tools/profile_pmc.sh" % frames_per_launch,
           "bytes_per_frame": t}, open(out, "w"), indent=1)
