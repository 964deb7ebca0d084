#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv rows per kernel and counter.
    python tools/pmc_summary.py gpurun_out/pmc_x [...more dirs]
"""
import csv
import glob
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def main():
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for d in sys.argv[1:]:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add((path, row["Dispatch_Id"]))
        for path in glob.glob(d + "/**/*.db", recursive=True):
            import sqlite3
            con = sqlite3.connect(path)
            for name, disp, cname, val in con.execute(
                    "select kernel_name, dispatch_id, counter_name, value from counters_collection"):
                k = short(name)
                tot[k][cname] += float(val)
                calls[k].add((path, disp))
    counters = sorted({c for k in tot for c in tot[k]})
    print("kernel".ljust(62), "calls".rjust(6), *(c[-22:].rjust(23) for c in counters))
    for k in sorted(tot, key=lambda k: -max(tot[k].values())):
        n = len(calls[k])
        print(k.ljust(62), str(n).rjust(6), *(f"{tot[k].get(c, 0) / n:23.4g}" for c in counters))


if __name__ == "__main__":
    main()
