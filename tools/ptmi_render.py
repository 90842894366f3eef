#!/usr/bin/env python3
"""Command-line caller of the C ABI (SURVEY 8b lists a CLI among the callers; the reference itself has none, its
`main` ignores argv): load a scene, optionally run the radiosity pre-pass, render one frame, save a PNG.

  python tools/ptmi_render.py --scene tests/golden/scenes/cbox.obj --width 512 --height 512 --spp 64 --out cbox.png
  python tools/ptmi_render.py --scene ... --subdivision 2 --radiosity --sampling-mode 3 --out guided.png
  python tools/ptmi_render.py --scene ... --radiosity --integrator radiosity --out radiosity_view.png
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-pathtracer_amd", "python"))
import ptmi  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scene", required=True)
    ap.add_argument("--width", type=int, default=800); ap.add_argument("--height", type=int, default=800)   # DEFAULT_WIDTH/HEIGHT
    ap.add_argument("--spp", type=int, default=16); ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--seed-base", type=int, default=2023)
    ap.add_argument("--subdivision", type=int, default=0); ap.add_argument("--convert-quads", action="store_true")
    ap.add_argument("--sampling-mode", type=int, default=0, help="0 BSDF, 1/2/4 grid, 3 MIS (render_config.h:38-44)")
    ap.add_argument("--mis-bsdf-fraction", type=float, default=0.5)
    ap.add_argument("--integrator", choices=["path", "radiosity"], default="path")
    ap.add_argument("--radiosity", action="store_true", help="run the radiosity pre-pass first (needed by guided modes / the radiosity view)")
    ap.add_argument("--radiosity-steps", type=int, default=10); ap.add_argument("--mc-samples", type=int, default=64)
    ap.add_argument("--point-to-point", action="store_true")
    ap.add_argument("--filter", choices=["none", "bilateral", "gaussian"], default="none", help="Apply Filter & Rebuild CDFs")
    ap.add_argument("--yaw", type=float, default=None); ap.add_argument("--pitch", type=float, default=None); ap.add_argument("--fov", type=float, default=None)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None, help="PNG file (top row first, like the reference's Save PNG)")
    a = ap.parse_args()

    r = ptmi.Renderer(a.device)
    t = time.time()
    r.load_scene(a.scene, a.subdivision, a.convert_quads)
    info = r.scene_info()
    print(f"scene: {info['n_prims']} primitives ({info['n_tris']} triangles, {info['n_quads']} quads), "
          f"{info['n_bvh_nodes']} BVH nodes, depth {info['bvh_depth']}  [{time.time() - t:.2f} s]")
    if a.radiosity:
        st = r.run_radiosity_solver(num_iterations=a.radiosity_steps, mc_samples=a.mc_samples, use_monte_carlo=not a.point_to_point)
        print(f"radiosity: {st.pairs} pairs, {st.rays} shadow rays, form factors {st.form_factor_ms:.1f} ms, "
              f"iterations {st.iteration_ms:.1f} ms, grids {st.grid_ms:.1f} ms")
        if a.filter != "none":
            r.apply_grid_filter(a.filter == "bilateral")
    cam = ptmi.default_camera()
    if a.yaw is not None: cam.yaw_deg = a.yaw
    if a.pitch is not None: cam.pitch_deg = a.pitch
    if a.fov is not None: cam.vfov_deg = a.fov
    r.set_camera(cam)
    r.update_resolution(a.width, a.height)
    r.set_config(spp=a.spp, max_depth=a.max_depth, seed_base=a.seed_base, sampling_mode=a.sampling_mode,
                 mis_bsdf_fraction=a.mis_bsdf_fraction, integrator=1 if a.integrator == "radiosity" else 0)
    st = r.render_frame()
    print(f"frame: {a.width}x{a.height} x {a.spp} spp in {st.seconds * 1e3:.2f} ms = {st.samples / st.seconds / 1e6:.1f} Msamples/s")
    if a.out:
        rgb, _ = r.read_image()
        ptmi.write_png(a.out, rgb)
        print(f"wrote {a.out}")
    r.close()


if __name__ == "__main__":
    main()
