#!/usr/bin/env python3
"""Per-kernel statistics (calls, total, average, share) from a rocprofv3 --kernel-trace results .db or
*_kernel_stats.csv; prints CSV.    python tools/kernel_stats.py gpurun_out/prof_x > profiles/x.csv"""
import glob
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    tot = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for d in sys.argv[1:]:
        for path in glob.glob(d + "/**/*.db", recursive=True):
            con = sqlite3.connect(path)
            for name, dur in con.execute("select name, duration from kernels"):
                t = tot[short(name)]
                t[0] += 1; t[1] += dur; t[2] = min(t[2], dur); t[3] = max(t[3], dur)
    total = sum(t[1] for t in tot.values())
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
    for k, t in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f'"{k}",{t[0]},{int(t[1])},{t[1] / t[0]:.1f},{100 * t[1] / total:.2f},{int(t[2])},{int(t[3])}')


if __name__ == "__main__":
    main()
