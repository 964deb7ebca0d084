#!/usr/bin/env python3
"""Copies what tools/profile_all.sh left under gpurun_out/prof_all into profiles/ (r03_* names) and checks that the
counter passes carry the stamp of the kernel sources in this tree."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.api import source_stamp  # noqa: E402

src = os.path.join(ROOT, "gpurun_out", "prof_all")
dst = os.path.join(ROOT, "profiles")
for w in ("cfg1", "cfg2", "cfg3", "cfg4"):
    shutil.copy(os.path.join(src, f"{w}_pmc.json"), os.path.join(dst, f"r03_pmc_{w}.json"))
    shutil.copy(os.path.join(src, f"{w}_kernel_stats.csv"), os.path.join(dst, f"r03_kernel_stats_{w}.csv"))
shutil.copy(os.path.join(src, "cfg5_kernel_stats.csv"), os.path.join(dst, "r03_kernel_stats_cfg5.csv"))
out = ("HIP API calls of tools/stop_motion_time.py (10 000 photons, 512^2; per frame: new grid -> pb_set_grid, pb_recast_beams, "
       "pb_build_lbvh, one heterogeneous frame with download) under rocprofv3 --hip-trace --stats, for 10 and for 30 timed frames "
       "(3 warm-up frames each).  Allocator calls do not grow with the frame count: the steady state allocates nothing.\n")
for n in (10, 30):
    line = [ln for ln in open(os.path.join(src, f"sm_{n}.log")) if "photons" in ln][-1].strip()
    out += f"\n== {n} frames: {line}\n"
    for ln in open(os.path.join(src, f"sm_hip_{n}", "sm_hip_api_stats.csv")):
        if re.search(r"Name|[Mm]alloc|[Ff]ree|Synchronize|hipLaunchKernel|hipMemcpy", ln):
            out += ",".join(ln.strip().split(",")[:4]) + "\n"
open(os.path.join(dst, "r03_stop_motion_hip_api.txt"), "w").write(out)
cur = source_stamp()
ok = True
for w in ("cfg1", "cfg2", "cfg3", "cfg4"):
    st = json.load(open(os.path.join(dst, f"r03_pmc_{w}.json")))["source_stamp"]
    print(w, st, "current" if st == cur else f"STALE (tree is {cur})")
    ok = ok and st == cur
sys.exit(0 if ok else 1)
