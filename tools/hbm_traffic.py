#!/usr/bin/env python3
"""HBM bytes per frame from two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) over bench.py.
    python tools/hbm_traffic.py <fetch_dir> <write_dir> <workload> > profiles/r01_hbm_traffic_<workload>.json
Counters are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes).  Kernels of the
counted (untimed) frame and of the one-time passes are left out."""
import glob
import json
import sqlite3
import sys
from collections import defaultdict

FRAME_KERNELS = ("setup_count_kernel", "queue_build_kernel", "fill_items_kernel", "eval_bins_fast_kernel", "finalize_kernel",
                 "eval_bins_kernel<false, true, 73>", "ray_setup_kernel<false>", "bin_kernel<false, true>",
                 "bin_kernel<true, true>", "make_items_kernel")


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def per_kernel(d, counter):
    tot, calls = defaultdict(float), defaultdict(set)
    for path in glob.glob(d + "/**/*.db", recursive=True):
        con = sqlite3.connect(path)
        for name, disp, cname, val in con.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            if cname == counter:
                tot[short(name)] += float(val)
                calls[short(name)].add(disp)
    return {k: tot[k] / len(calls[k]) for k in tot}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    shared = "setup_count_kernel" in fetch or "setup_count_kernel" in write
    separate = ("ray_setup_kernel<false>", "bin_kernel<false, true>", "bin_kernel<true, true>", "make_items_kernel")
    for k in FRAME_KERNELS:
        if shared and k in separate:
            continue          # launched by the counted (untimed) frame only
        if k in fetch or k in write:
            kernels[k] = {"fetch_bytes": 2 * 1024 * fetch.get(k, 0.0), "write_bytes": 1024 * write.get(k, 0.0)}
    f = sum(v["fetch_bytes"] for v in kernels.values()); w = sum(v["write_bytes"] for v in kernels.values())
    print(json.dumps({"workload": sys.argv[3], "cull": "bins", "unit": "bytes per frame (one launch of each frame kernel)",
                      "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 10 "
                                "--warmup 2 --no-cpu-baseline --in-flight 0`; counters are in KiB; FETCH_SIZE doubled (gfx950 "
                                "tallies 128-B requests at 64 B)",
                      "kernels": kernels, "fetch_bytes": f, "write_bytes": w, "traffic_bytes": f + w}, indent=1))


if __name__ == "__main__":
    main()
