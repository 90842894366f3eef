"""Writes the committed golden frames (tests/golden/frame_*.npz) from the CPU oracle.

Provenance: the oracle (oracle/ptmi_oracle.c) is a restatement whose geometry layer is checked bit-for-bit
against the reference's own headers compiled from /root/reference (tests/test_oracle_vs_ref.py) and whose
loader/BVH/camera are checked against the known answers in SURVEY.md §8c; the integrator loop and the
cuRAND XORWOW restatement are NOT pinned by any reference execution (the reference ships no tests or
fixtures for this path and its integrator.h cannot be compiled here).  These files therefore pin the
ORACLE (regression) and give the GPU tests a fixture that does not need the oracle library.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_binding import OracleScene, SCENES, default_camera  # noqa: E402

CASES = [  # scene, subdivision, convert_quads, W, H, spp, max_depth
    ("cbox.obj", 0, False, 64, 64, 4, 5),
    ("cbox.obj", 0, False, 128, 128, 16, 4),
    ("cbox.obj", 0, False, 64, 64, 16, 8),
    ("cbox_quads.obj", 0, False, 96, 54, 16, 5),
    ("cbox_quads.obj", 0, True, 64, 64, 8, 5),
]

for scene, sub, conv, W, H, spp, depth in CASES:
    o = OracleScene.load(os.path.join(SCENES, scene), sub, conv)
    rgb, rad, st = o.render(default_camera(), W, H, spp, max_depth=depth)
    name = f"frame_{scene.split('.')[0]}_s{sub}c{int(conv)}_{W}x{H}_{spp}spp_d{depth}.npz"
    np.savez_compressed(os.path.join(HERE, name), scene=scene, subdivision=sub, convert_quads=conv, width=W, height=H,
                        spp=spp, max_depth=depth, rgb8=rgb, radiance=rad,
                        counters=np.array([st.samples, st.rays, st.node_visits, st.prim_tests, st.hits], np.uint64))
    print(name, float(rad.mean()), st.rays / st.samples)
