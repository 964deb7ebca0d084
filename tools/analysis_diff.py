"""Cells of the GPU's NSD.csv / rRMSE.csv that differ as text from the golden files."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
import raytracer_amd as pb
with pb.Context(pb.control_defaults()) as ctx:
    for table, ref in ((0, "reference_NSD.csv"), (1, "ref_rRMSE.csv")):
        mine = orc.analysis_csv(*ctx.run_analysis(table)).splitlines()
        gold = open(os.path.join(ROOT, "tests", "golden", ref)).read().splitlines()
        cells = sum(len(l.split(",")) - 1 for l in gold)
        diff = sum(x != y for a, b in zip(mine, gold) for x, y in zip(a.split(","), b.split(",")))
        print(ref, "rows", len(mine), len(gold), "cells", cells, "differ", diff)
