#!/usr/bin/env python
"""Development tool (MI355X box): the bench's hard-move legs on their own (bench.hard_moves_leg), optionally with the
QPBO solver's own phase report (STEREO_HIP_QPBO_VERBOSE=1).   tools/time_hard_moves.py [teddy|synthetic] [noref]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "teddy"
if len(sys.argv) > 2 and sys.argv[2] == "noref":
    from oracle import pyoracle
    pyoracle.have_ref_qpbo = lambda: False
im0, _ = bench.synthetic_pair(375, 450, 60)
print(json.dumps(bench.hard_moves_leg(which, 375, 450, im0), indent=1))
