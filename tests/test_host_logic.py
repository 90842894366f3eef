"""CPU-only tests of the PRODUCT's host half (libptmi.so through the C ABI): the C++ loader, quad
conversion, subdivision, BVH builder, camera and tiling arithmetic must agree bit-for-bit with the
oracle (an independent plain-C restatement).  No GPU is touched; no render call is made."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ptmi
from oracle_binding import OracleScene, SCENES, Camera as OCamera, camera_frame, default_camera, ref_obj_available, ref_obj_load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def bits(a):
    return np.ascontiguousarray(a, F).view(np.uint32)


def test_library_loads_and_exports_every_declared_symbol():
    L = ptmi.lib()
    header = open(os.path.join(ROOT, "include", "ptmi.h")).read()
    declared = set(re.findall(r"\b(ptmi_[a-z0-9_]+)\s*\(", header))
    declared -= {"ptmi_tiling"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(L, name), f"libptmi.so does not export {name}"
    assert declared == set(ptmi.EXPORTS), declared ^ set(ptmi.EXPORTS)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ptmi.PtmiError) as e:
        ptmi.Renderer(0)
    assert e.value.code == -3


VARIANTS = [("cbox.obj", 0, False), ("cbox_quads.obj", 0, False), ("cbox_quads.obj", 0, True), ("cbox.obj", 1, False),
            ("cbox.obj", 2, False), ("cbox_quads.obj", 1, False), ("cbox_quads.obj", 2, True), ("cbox_quads.obj", 3, False)]


@pytest.mark.parametrize("name,sub,conv", VARIANTS)
def test_loader_and_bvh_match_oracle(name, sub, conv):
    path = os.path.join(SCENES, name)
    h = ptmi.HostScene.load(path, sub, conv)
    o = OracleScene.load(path, sub, conv)
    hp, op = h.prims(), o.prims()
    assert h.info()["n_prims"] == o.n_prims and h.info()["n_bvh_nodes"] == o.n_nodes
    assert (hp["type"] == op["type"]).all()
    tri = hp["type"] == 0
    assert (bits(hp["verts"][tri][:, :3]) == bits(op["verts"][tri][:, :3])).all()
    assert (bits(hp["verts"][~tri]) == bits(op["verts"][~tri])).all()
    for k in ("normal", "bsdf", "Le"):
        assert (bits(hp[k]) == bits(op[k])).all(), k
    hb, ob = h.bvh(), o.bvh()
    for k in ("left", "count", "indices"):
        assert (hb[k] == ob[k]).all(), k
    inner = hb["count"] == 0
    assert (hb["right"][inner] == ob["right"][inner]).all()
    assert (bits(hb["bmin"]) == bits(ob["bmin"])).all() and (bits(hb["bmax"]) == bits(ob["bmax"])).all()


def test_arrays_scene_and_degenerate_bvh():
    rng = np.random.default_rng(3)
    n = 3000
    types = (rng.random(n) < 0.4).astype(np.int32)
    verts = (rng.uniform(-5, 5, (n, 1, 3)) + rng.normal(0, 0.3, (n, 4, 3))).astype(F)
    normal = rng.normal(0, 1, (n, 3)).astype(F); bsdf = rng.random((n, 3)).astype(F); Le = np.zeros((n, 3), F)
    # a clump of identical primitives forces the "centroid extent < 1e-6" oversized leaf
    verts[100:120] = verts[100]; types[100:120] = types[100]
    h = ptmi.HostScene.from_arrays(types, verts, normal, bsdf, Le)
    o = OracleScene.from_arrays(types, verts, normal, bsdf, Le)
    hb, ob = h.bvh(), o.bvh()
    assert len(hb["left"]) == len(ob["left"])
    for k in ("left", "count", "indices"):
        assert (hb[k] == ob[k]).all(), k
    assert (bits(hb["bmin"]) == bits(ob["bmin"])).all()
    assert hb["count"].max() > 4
    assert h.info()["bvh_depth"] >= 8


def test_loader_error_behaviour(tmp_path):
    with pytest.raises(ptmi.PtmiError) as e:
        ptmi.HostScene.load(str(tmp_path / "missing.obj"))
    assert e.value.code == -2
    p = tmp_path / "scene.ply"; p.write_text("ply\n")
    with pytest.raises(ptmi.PtmiError) as e:
        ptmi.HostScene.load(str(p))
    assert e.value.code == -2 and "unsupported" in str(e.value)
    p = tmp_path / "empty.obj"; p.write_text("# nothing\nv 0 0 0\n")
    with pytest.raises(ptmi.PtmiError):
        ptmi.HostScene.load(str(p))


def test_loader_quirks(tmp_path):
    """Ragged input the reference loader tolerates (file_manager.h:118-250)."""
    (tmp_path / "m.mtl").write_text("newmtl A\nKd 0.1 0.2 0.3\nKe 1 2 3\nKs 9 9 9\n\nnewmtl B\n  Kd 0.5 0.5 0.5\n")
    obj = tmp_path / "q.OBJ"          # extension check is case-insensitive
    obj.write_text("\n".join([
        "mtllib m.mtl", "v 0 0 0", "v 1 0 0", "v 1 1 0", "v 0 1 0", "v 0 0 1", "vn 0 0 2", "vt 0.5 0.5",
        "object_line_is_skipped_because_it_starts_with_o", "s off",
        "usemtl A", "f 1/1/1 2/1/1 3/1/1", "f 1//1 2//1 3//1 4//1   # trailing comment tokens are skipped",
        "usemtl NOPE", "f 1 2 5", "f 1 2", "f 1 2 3 4 5", "f 1 2 99", "f -1 2 3", "usemtl B", "f 1/7 2/7 3/7", ""]))
    h = ptmi.HostScene.load(str(obj)); o = OracleScene.load(str(obj))
    hp, op = h.prims(), o.prims()
    assert hp["type"].tolist() == [0, 1, 0, 0] == op["type"].tolist()
    assert np.allclose(hp["normal"][0], [0, 0, 1]) and np.allclose(hp["Le"][0], [1, 2, 3]) and np.allclose(hp["bsdf"][0], [0.1, 0.2, 0.3])
    assert np.allclose(hp["bsdf"][2], [0.8, 0.8, 0.8]) and np.allclose(hp["Le"][2], 0)      # unknown material -> default
    assert np.allclose(hp["bsdf"][3], [0.5, 0.5, 0.5])
    for k in ("normal", "bsdf", "Le"):
        assert (bits(hp[k]) == bits(op[k])).all()


def test_camera_frame_matches_oracle():
    cams = [ptmi.default_camera(),
            ptmi.Camera((0.5, 3.0, 8.5), (0, 2.5, 0), (0, 1, 0), 40.0, 37.5, -12.25, 1),
            ptmi.Camera((1.0, 2.0, 7.0), (0.2, 2.5, -1), (0, 1, 0), 30.0, 90.0, 0.0, 0),
            ptmi.Camera((-2.0, 4.0, 3.0), (0, 2.5, -3), (0.1, 1, 0), 70.0, 200.0, -45.0, 1)]
    for cam in cams:
        ocam = OCamera(tuple(cam.origin), tuple(cam.lookat), tuple(cam.vup), cam.vfov_deg, cam.yaw_deg, cam.pitch_deg, cam.orbit)
        for (w, hgt) in ((1024, 1024), (1920, 1080), (200, 333)):
            assert (bits(ptmi.host_camera_frame(cam, w, hgt)) == bits(camera_frame(ocam, w, hgt).as_array())).all()


def test_tiling_partitions_rows_exactly_once():
    for height in (1, 7, 64, 1080, 1448):
        for n_ranks in (1, 2, 3, 4, 8):
            for row_block in (1, 4, 8, 16):
                seen = np.concatenate([ptmi.host_local_row_map(height, n_ranks, r, row_block) for r in range(n_ranks)])
                assert sorted(seen.tolist()) == list(range(height))
                for r in range(n_ranks):
                    rows = ptmi.host_local_row_map(height, n_ranks, r, row_block)
                    assert ((rows // row_block) % n_ranks == r).all() and (np.diff(rows) > 0).all()
    with pytest.raises(ptmi.PtmiError):
        ptmi.host_local_row_map(16, 2, 2, 8)


def test_c_abi_from_plain_c(tmp_path):
    """include/ptmi.h must be usable from C99 and link against libptmi.so (examples/host_only.c)."""
    import subprocess
    exe = tmp_path / "host_only"
    libdir = os.path.join(ROOT, "cuda-pathtracer_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "host_only.c"), "-L" + libdir, "-lptmi",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    out = subprocess.run([str(exe), os.path.join(SCENES, "cbox.obj")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "prims 32 (tris 32 quads 0) bvh nodes 21 depth" in out.stdout
    assert "origin -3.72830215e-07 2.5 8.52936077" in out.stdout
    assert "rank 3 of 8 owns 512 of 4096 rows" in out.stdout


def test_png_writer_round_trip(tmp_path):
    """"Save PNG" (ui_windows.h:195-210): 8-bit RGB, the frame's bottom row (row 0 of the image buffer) last in the file.
    Decoded here with zlib only: signature, IHDR, CRCs, filter bytes, pixels."""
    import struct, zlib
    rng = np.random.default_rng(4)
    for (h, w) in [(1, 1), (7, 5), (300, 211)]:                     # the last one needs more than one stored deflate block
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        path = str(tmp_path / f"t{h}.png")
        ptmi.write_png(path, img)
        data = open(path, "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        pos = 8; chunks = []
        while pos < len(data):
            n, typ = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
            assert crc == zlib.crc32(typ + body) & 0xffffffff
            chunks.append((typ, body)); pos += 12 + n
        assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
        assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
        raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)
        assert (raw[:, 0] == 0).all()
        assert (raw[:, 1:].reshape(h, w, 3) == img[::-1]).all()       # flipped: top row first
    with pytest.raises(ptmi.PtmiError):
        ptmi.write_png(str(tmp_path / "no_such_dir" / "x.png"), img)


def test_loader_differential_fuzz(tmp_path):
    """300 random OBJ/MTL files from a grammar of everything the reference loader reacts to (and much it ignores):
    the product's C++ loader (+ conversion, subdivision, BVH) and the oracle's C restatement - written independently
    from file_manager.h:39-273 - must agree on accept/reject and, where accepted, on every bit."""
    rng = np.random.default_rng(2024)
    num = lambda: rng.choice([f"{rng.uniform(-4, 4):.4f}", f"{rng.integers(-3, 4)}", f"{rng.uniform(-1, 1):.3e}", "0", "-0.0", "1e-3", ".5", "+2"])
    agree_ok = agree_fail = ref_ok = ref_fail = 0
    for case in range(300):
        nv = int(rng.integers(0, 9)) if rng.random() < 0.15 else int(rng.integers(3, 10)); nn = int(rng.integers(0, 4))
        mtl_lines = []
        for m in range(int(rng.integers(0, 4))):
            mtl_lines.append(f"newmtl M{m}")
            for _ in range(int(rng.integers(0, 4))):
                mtl_lines.append(rng.choice(["Kd", "Ke", "Ks", "Ka", "  Kd", "\tKe", "Ns", "# c", "d"]) + " " + " ".join(num() for _ in range(int(rng.integers(0, 5)))))
        lines = []
        if rng.random() < 0.8: lines.append(rng.choice(["mtllib m.mtl", "mtllib missing.mtl", "mtllib"]))
        for _ in range(nv): lines.append("v " + " ".join(num() for _ in range(int(rng.choice([3, 3, 3, 3, 3, 3, 3, 2, 4])))))
        for _ in range(nn): lines.append("vn " + " ".join(num() for _ in range(3)))
        for _ in range(int(rng.integers(0, 10))):
            kind = rng.random()
            if kind < 0.15: lines.append(f"usemtl M{rng.integers(0, 5)}")
            elif kind < 0.25: lines.append(rng.choice(["", "# comment", "o thing", "g grp", "s 1", "vt 0 1", "   ", "f", "v"]))
            else:
                k = int(rng.choice([1, 2, 3, 3, 3, 4, 4, 5]))
                toks = []
                for _ in range(k):
                    vi = int(rng.integers(1, nv + 1)) if (nv and rng.random() < 0.93) else int(rng.integers(-1, nv + 3))
                    form = rng.random()
                    if form < 0.4: toks.append(f"{vi}")
                    elif form < 0.6: toks.append(f"{vi}/{rng.integers(0, 3)}")
                    elif form < 0.85: toks.append(f"{vi}//{rng.integers(0, nn + 2)}")
                    else: toks.append(f"{vi}/{rng.integers(0, 3)}/{rng.integers(0, nn + 2)}")
                if rng.random() < 0.1: toks.append("# tail")
                lines.append("f " + " ".join(toks))
        d = tmp_path / f"c{case}"; d.mkdir()
        (d / "m.mtl").write_text("\n".join(mtl_lines) + "\n")
        (d / "s.obj").write_text("\n".join(lines) + ("\n" if rng.random() < 0.9 else ""))
        sub = int(rng.integers(0, 3)); conv = bool(rng.random() < 0.5)
        # a Kd / Ke line with fewer than three numbers makes the reference read uninitialised floats (file_manager.h:62-69:
        # `float r, g, b; iss >> r >> g >> b;` unchecked - in the compiled reference the previous line's values show through);
        # both restatements store 0 for what is missing, and such files are left out of the comparison with the compiled reference
        short_colour = any(l.split()[:1] in (["Kd"], ["Ke"]) and len(l.split()) < 4 for l in mtl_lines)
        if ref_obj_available() and not short_colour:
            # the reference's own loadOBJ / loadMTL (utils/file_manager.h:39-79, 93-273, compiled into oracle/_ref): the product's
            # loader and the oracle's must take its decision and reproduce every bit of what it returns
            want = ref_obj_load(str(d / "s.obj"))
            got = []
            for load in (ptmi.HostScene.load, OracleScene.load):
                try:
                    got.append(load(str(d / "s.obj"), 0, False).prims())
                except Exception:
                    got.append(None)
            for g in got:
                assert (g is None) == (want is None or len(want["type"]) == 0), (case, lines)
                if g is not None:
                    assert (g["type"] == want["type"]).all(), case
                    tri = g["type"] == 0
                    assert (bits(g["verts"][tri][:, :3]) == bits(want["verts"][tri][:, :3])).all() and (bits(g["verts"][~tri]) == bits(want["verts"][~tri])).all(), case
                    for k in ("normal", "bsdf", "Le"):
                        assert (bits(g[k]) == bits(want[k])).all(), (case, k)
            ref_ok += got[0] is not None; ref_fail += got[0] is None
        try:
            h = ptmi.HostScene.load(str(d / "s.obj"), sub, conv)
        except ptmi.PtmiError:
            h = None
        o = None
        try:
            o = OracleScene.load(str(d / "s.obj"), sub, conv)
        except Exception:
            o = None
        assert (h is None) == (o is None), (case, lines)
        if h is None:
            agree_fail += 1
            continue
        agree_ok += 1
        hp, op = h.prims(), o.prims()
        assert (hp["type"] == op["type"]).all(), case
        tri = hp["type"] == 0
        assert (bits(hp["verts"][tri][:, :3]) == bits(op["verts"][tri][:, :3])).all() and (bits(hp["verts"][~tri]) == bits(op["verts"][~tri])).all(), case
        for k in ("normal", "bsdf", "Le"):
            assert (bits(hp[k]) == bits(op[k])).all(), (case, k)
        hb, ob = h.bvh(), o.bvh()
        assert len(hb["left"]) == len(ob["left"]) and (hb["indices"] == ob["indices"]).all() and (hb["count"] == ob["count"]).all(), case
        assert (bits(hb["bmin"]) == bits(ob["bmin"])).all() and (bits(hb["bmax"]) == bits(ob["bmax"])).all(), case
    assert agree_ok > 60 and agree_fail > 20, (agree_ok, agree_fail)
    print(f"loader fuzz: {agree_ok} accepted, {agree_fail} rejected, all agreeing; against the compiled reference loader: {ref_ok} accepted, {ref_fail} rejected")
    if ref_obj_available():
        assert ref_ok > 60 and ref_fail > 20
