"""Writes the committed golden frames (tests/golden/frame_*.npz) from the CPU oracle.

Provenance: the oracle (oracle/ptmi_oracle.c) is a restatement whose geometry layer is checked bit-for-bit
against the reference's own headers compiled from /root/reference (tests/test_oracle_vs_ref.py) and whose
loader/BVH/camera are checked against the known answers in SURVEY.md §8c; the integrator loop and the
cuRAND XORWOW restatement are NOT pinned by any reference execution (the reference ships no tests or
fixtures for this path and its integrator.h cannot be compiled here).  These files therefore pin the
ORACLE (regression) and give the GPU tests a fixture that does not need the oracle library.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
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
    ("cbox.obj", 0, False, 256, 256, 16, 4),          # BASELINE.json configs[0] (C1) at its stated size
]

for scene, sub, conv, W, H, spp, depth in CASES if "--solver-only" not in sys.argv else []:
    o = OracleScene.load(os.path.join(SCENES, scene), sub, conv)
    rgb, rad, st = o.render(default_camera(), W, H, spp, max_depth=depth)
    name = f"frame_{scene.split('.')[0]}_s{sub}c{int(conv)}_{W}x{H}_{spp}spp_d{depth}.npz"
    np.savez_compressed(os.path.join(HERE, name), scene=scene, subdivision=sub, convert_quads=conv, width=W, height=H,
                        spp=spp, max_depth=depth, rgb8=rgb, radiance=rad,
                        counters=np.array([st.samples, st.rays, st.node_visits, st.prim_tests, st.hits], np.uint64))
    print(name, float(rad.mean()), st.rays / st.samples)


# radiosity pre-pass + what it feeds: solver outputs, a MIS-guided frame and a radiosity view
SOLVER_CASES = [  # scene, subdivision, solver parameters
    ("cbox.obj", 0, dict()),                                                     # RadiosityState defaults: MC, 64 samples, 10 steps
    ("cbox_quads.obj", 1, dict(mc_samples=16, num_iterations=3, enable_filtering=True)),
    ("cbox.obj", 1, dict(use_monte_carlo=False, num_iterations=5)),
]
for scene, sub, prm in SOLVER_CASES:
    o = OracleScene.load(os.path.join(SCENES, scene), sub, False)
    sol = o.radiosity_solve(**prm)
    W, H = 48, 40
    state = np.zeros((H * W, 6), np.uint32)
    g_rgb, g_rad, _ = o.render(default_camera(), W, H, 4, sampling_mode=3, rng_state=state)
    v_rgb, v_rad = o.render_radiosity(default_camera(), W, H, 2, rng_state=state, reset_rng=False)
    name = f"solver_{scene.split('.')[0]}_s{sub}_" + ("default" if not prm else "_".join(f"{k}{int(v)}" for k, v in prm.items())) + ".npz"
    np.savez_compressed(os.path.join(HERE, name), scene=scene, subdivision=sub, params=json.dumps(prm), width=W, height=H,
                        form_factors=sol["form_factors"], radiosity=sol["radiosity"], unshot=sol["unshot"], grid=sol["grid"],
                        radiosity_grid=sol["radiosity_grid"], rays=np.uint64(sol["rays"]), cdfs=o.cdfs(),
                        guided_rgb8=g_rgb, guided_radiance=g_rad, view_rgb8=v_rgb, view_radiance=v_rad)
    print(name, float(sol["form_factors"].sum()), float(sol["radiosity"].mean()), sol["rays"])


# PBRT import: the primitive arrays the REFERENCE's own loadPBRT (utils/pbrt_loader.h:178-422, compiled with the
# vendored pbrtParser into oracle/_ref/libptmi_ref_pbrt.so by oracle/Makefile) returns for the fixtures under golden/pbrt/.
# Only where /root/reference exists; the .npz files travel.  A fixture the reference rejects gets failed = True.
from oracle_binding import ref_pbrt_available, ref_pbrt_load  # noqa: E402
import glob  # noqa: E402
if ref_pbrt_available():
    for f in sorted(glob.glob(os.path.join(HERE, "pbrt", "*.pbrt"))):
        got = ref_pbrt_load(f)
        out = os.path.splitext(f)[0] + ".npz"
        if got is None:
            np.savez_compressed(out, failed=True)
        else:
            np.savez_compressed(out, failed=False, **got)
        print(os.path.basename(out), "rejected" if got is None else len(got["type"]))
