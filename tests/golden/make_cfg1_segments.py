#!/usr/bin/env python3
"""Writes tests/golden/cfg1_tree_segments.bin: the 987 segments the reference's tree holds at Control.h as
committed (1000 photons, SEED 42, std::mt19937), each with its leaf-node box as gate box and its energy as
BeamRadianceStandard looks it up, plus the root box, in the segment file format of include/photon_beams.h.
Source: the CPU oracle's CastPhotonBeams -> BoundPhotonBeams -> BuildTree (oracle/pb_oracle.cpp), the run
that reproduces the reference's committed screenshot and its printed counts (tests/test_oracle_golden.py).
Needs no GPU:    python tests/golden/make_cfg1_segments.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
import raytracer_amd as pb  # noqa: E402

o = orc.Oracle()
assert o.cast_photon_beams() == 957 and o.bound_photon_beams() == 1059 and o.build_tree() == 987
segs = o.export_segments(tree=True)
lo, hi = o.root_box()
pb.write_segments_file(os.path.join(HERE, "cfg1_tree_segments.bin"), segs, lo, hi)
print("wrote", len(segs), "segments")
