"""Timing probe for the radiosity pre-pass: python tools/radiosity_probe.py [sub ...] [--oracle] [--p2p] [--samples N] [--fast | --walk W]
   --fast : the visibility walk through the opt-in fast tree, no certificate;  --walk 0 : the reference's own walk, --walk 2 : certified
           (default: automatic = certified from 256 triangles up)
   --check-profile FILE : only check that FILE (profiles/rNN_pmc_radiosity.json) was taken from the solver kernels that are built
                          now (stamps of ptmi_buildinfo: the library, or the sources compiled into build/radiosity.o); exit 1 if not"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "cuda-pathtracer_amd", "python"), os.path.join(ROOT, "tests")]
import ptmi

if "--check-profile" in sys.argv:
    import json, ptmi_buildinfo
    path = sys.argv[sys.argv.index("--check-profile") + 1]
    ok, how = ptmi_buildinfo.profile_is_current(json.load(open(path)), solver=True)
    print(f"{path}: {'current (matched by ' + how + ')' if ok else 'STALE: taken from other solver kernels than the ones built now - re-take it (tools/profile_rad.sh)'}")
    sys.exit(0 if ok else 1)

args = [a for a in sys.argv[1:] if not a.startswith("--") and a.isdigit()]
subs = [int(a) for a in args] or [2, 3, 4]
want_oracle = "--oracle" in sys.argv
kw = {}
if "--p2p" in sys.argv: kw["use_monte_carlo"] = 0
if "--samples" in sys.argv: kw["mc_samples"] = int(sys.argv[sys.argv.index("--samples") + 1])
fast = "--fast" in sys.argv            # the visibility walk through the opt-in fast tree (ptmi_config.fast_tree)
scene = os.path.join(ROOT, "tests", "golden", "scenes", "cbox.obj")
R = ptmi.Renderer(0)
R.set_config(fast_tree=fast)
if "--walk" in sys.argv: R.set_solver_walk(int(sys.argv[sys.argv.index("--walk") + 1]))
for sub in subs:
    R.load_scene(scene, sub, False)
    n = R.scene_info()["n_prims"]
    for rep in range(2):
        t = time.time(); st = R.run_radiosity_solver(**kw); wall = time.time() - t
    print(f"sub {sub}: n={n} walk={st.walk} chains={st.cert_chain} fallbacks={st.cert_fallback} pairs={st.pairs} rays={st.rays} | form factors {st.form_factor_ms:.2f} ms "
          f"({st.rays / st.form_factor_ms / 1e3:.1f} Mrays/s, {st.pairs / st.form_factor_ms / 1e3:.1f} Mpairs/s) | "
          f"iterations {st.iteration_ms:.2f} ms | grids {st.grid_ms:.2f} ms | device {st.seconds * 1e3:.2f} ms | wall {wall * 1e3:.1f} ms", flush=True)
    if want_oracle:
        from oracle_binding import OracleScene
        o = OracleScene.load(scene, sub, False)
        t = time.time(); exp = o.radiosity_solve(**kw); dt = time.time() - t
        got = R.radiosity_solution()
        same = all((got[k].view(np.uint32) == exp[k].view(np.uint32)).all() for k in ("form_factors", "radiosity", "unshot", "grid", "radiosity_grid"))
        ndiff = int((got["form_factors"].view(np.uint32) != exp["form_factors"].view(np.uint32)).sum())
        print(f"   oracle {dt:.2f} s on {os.cpu_count()} threads ({exp['rays'] / dt / 1e6:.1f} Mrays/s); bit-identical: {same}; form factors that differ: {ndiff} of {n * n}, "
              f"rays {st.rays} vs {exp['rays']}", flush=True)
