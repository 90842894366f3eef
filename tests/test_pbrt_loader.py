"""PBRT import: the product's loader (host/pbrt_loader.cpp behind ptmi_load_scene / ptmi_host_scene_load)
against the REFERENCE's own loadPBRT - utils/pbrt_loader.h:178-422 compiled with the vendored pbrtParser into
oracle/_ref/libptmi_ref_pbrt.so - bit for bit on the primitive arrays:
  * committed goldens (tests/golden/pbrt/*.npz, written by make_golden.py from the compiled reference) - these run anywhere;
  * live, where oracle/_ref exists: the fixtures again, 300 random scenes (transform stacks incl. Rotate, instancing two
    levels deep, area-light scoping, every convertible material) and the > 2,000,000-triangle proxy path.
A file the reference rejects (returns false, throws, or would crash) must be rejected by the product with PTMI_E_IO."""
import glob
import os

import numpy as np
import pytest

import ptmi
from oracle_binding import ref_pbrt_available, ref_pbrt_load
from pbrt_fuzz import random_scene, random_scene_with_ply, random_scene_with_spd

HERE = os.path.dirname(os.path.abspath(__file__))
PBRT = os.path.join(HERE, "golden", "pbrt")
F = np.float32


def same(got, want, what):
    assert len(got["type"]) == len(want["type"]), (what, len(got["type"]), len(want["type"]))
    assert (got["type"] == want["type"]).all(), what
    for k in ("verts", "normal", "bsdf", "Le"):
        a, b = np.ascontiguousarray(got[k], F).view(np.uint32), np.ascontiguousarray(want[k], F).view(np.uint32)
        assert (a == b).all(), (what, k, np.argwhere(a != b)[:3].tolist())


def product_load(path):
    try:
        return ptmi.HostScene.load(path).prims()
    except ptmi.PtmiError as e:
        assert e.code == -2, e                       # PTMI_E_IO
        return None


def test_fixtures_match_the_committed_goldens():
    files = sorted(glob.glob(os.path.join(PBRT, "*.pbrt")))
    assert len(files) >= 12
    n_ok = n_rejected = 0
    for f in files:
        g = np.load(os.path.splitext(f)[0] + ".npz")
        got = product_load(f)
        if bool(g["failed"]):
            assert got is None, f"{os.path.basename(f)}: the reference rejects this file, the product loaded it"
            n_rejected += 1
        else:
            assert got is not None, os.path.basename(f)
            same(got, {k: g[k] for k in ("type", "verts", "normal", "bsdf", "Le")}, os.path.basename(f))
            assert (got["type"] == 0).all()              # the importer only ever makes triangles
            n_ok += 1
    assert n_ok >= 9 and n_rejected >= 9


def test_hand_checked_values():
    """what the first fixture must give, written out by hand: vertices as in the file, geometric normal, Kd as albedo"""
    p = product_load(os.path.join(PBRT, "01_trianglemesh.pbrt"))
    assert len(p["type"]) == 2
    assert p["verts"][0, :3].tolist() == [[-1, 0, -1], [1, 0, -1], [1, 0, 1]] and p["verts"][1, 2].tolist() == [-1, 0.25, 1]
    assert np.allclose(p["normal"][0], [0, -1, 0]) and np.allclose(p["bsdf"], [[0.725, 0.71, 0.68]] * 2) and not p["Le"].any()
    a = product_load(os.path.join(PBRT, "03_arealight.pbrt"))
    assert a["Le"].tolist() == [[17, 12, 4]] * 4 + [[0, 0, 0]]          # the FIRST area light of the scope wins; dropped at AttributeEnd
    m = product_load(os.path.join(PBRT, "04_materials.pbrt"))
    assert m["bsdf"][0].tolist() == [F(0.8)] * 3 and m["bsdf"][3].tolist() == [0, 0, 0]      # default material; metal: diffuse*(1-1) + 0*1
    assert np.allclose(m["bsdf"][10], [0.8 * 0.25 + 0.8 * 0.75 * 0.75, 0.3 * 0.25 + 0.3 * 0.75 * 0.75, 0.1 * 0.25 + 0.1 * 0.75 * 0.75])   # disney
    assert m["bsdf"][18].tolist() == [1, 1, 1]                            # a textured Kd counts as 1 1 1
    y = product_load(os.path.join(PBRT, "08_plymesh.pbrt"))                 # three .ply files: ASCII, binary LE (doubles), binary BE (CRLF)
    assert len(y["type"]) == 16
    assert y["verts"][0, :3].tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0.25]] and np.allclose(y["normal"][0], [0, 0, 1])   # the file's nx ny nz
    assert y["Le"][4:8].tolist() == [[4, 3, 2]] * 4 and not y["Le"][:4].any() and not y["Le"][8:].any()
    assert np.allclose(y["bsdf"][:4], [[0.3, 0.5, 0.7]] * 4) and np.allclose(y["verts"][8, 0], [-3, 0, 0])                 # instanced under Translate -3 0 0


def test_errors_and_unsupported_constructs(tmp_path):
    def rejected(text, name="x.pbrt"):
        f = tmp_path / name; f.write_text(text)
        return product_load(str(f)) is None
    tri = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\n'
    assert rejected("WorldBegin\n" + tri)                                                 # no WorldEnd: unexpected end of file
    assert rejected('WorldBegin\nShape "plymesh" "string filename" "m.ply"\nWorldEnd\n')  # no such file
    (tmp_path / "bad.ply").write_text("ply\nformat ascii 2.0\nelement vertex 0\nend_header\n")
    assert rejected('WorldBegin\nShape "plymesh" "string filename" "bad.ply"\n' + tri + "WorldEnd\n")   # header: not format 1.0
    (tmp_path / "quad.ply").write_text("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                                       "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    assert rejected('WorldBegin\nShape "plymesh" "string filename" "quad.ply"\nWorldEnd\n')   # triangles only, as in the library
    assert rejected('WorldBegin\nShape "cylinder" "float radius" [1]\n' + tri + "WorldEnd\n")   # the reference crashes on it
    assert rejected('WorldBegin\nObjectBegin "a"\n' + tri + 'ObjectInstance "a"\nObjectEnd\nObjectInstance "a"\nWorldEnd\n')   # recursion
    assert rejected('WorldBegin\nShape "trianglemesh" "integer indices" [0 1 7] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd\n')
    assert rejected('WorldBegin\nInclude "nosuchfile.pbrt"\n' + tri + "WorldEnd\n")
    assert rejected('Material "matte"\nWorldBegin\n' + tri + "WorldEnd\n")                # Material outside the world block
    # "spectrum" parameters from .spd files (Parser.inl:208-224): the file must be there, every token a number, pairs unless three
    (tmp_path / "ok.spd").write_text("# n\n400 0.2\n500 0.9\n")
    (tmp_path / "odd.spd").write_text("400 0.2 500 0.9 600\n")
    (tmp_path / "word.spd").write_text("400 0.2 nm 0.9\n")
    metal = lambda f: 'WorldBegin\nMaterial "metal" "spectrum eta" "' + f + '"\n' + tri + "WorldEnd\n"
    assert not rejected(metal("ok.spd")) and not rejected(metal(str(tmp_path / "ok.spd")))     # relative to the scene, or absolute
    assert rejected(metal("nosuch.spd")) and rejected(metal("odd.spd")) and rejected(metal("word.spd"))
    assert rejected('WorldBegin\nMaterial "matte" "spectrum Kd" "ok.spd"\n' + tri + "WorldEnd\n")      # four numbers are no RGB value
    assert not rejected("WorldBegin\n" + tri + "WorldEnd\n")
    with pytest.raises(ptmi.PtmiError):
        ptmi.HostScene.load(str(tmp_path / "missing.pbrt"))
    with pytest.raises(ptmi.PtmiError):
        ptmi.HostScene.load(str(tmp_path / "scene.usd"))


def test_pbrt_scene_goes_through_the_rest_of_load_scene():
    """loadScene's later stages apply to imported scenes as to OBJ ones: subdivision, BVH (application_state.h:404-438)"""
    hs = ptmi.HostScene.load(os.path.join(PBRT, "05_instances.pbrt"), subdivision_count=2)
    info = hs.info()
    assert info["n_prims"] == 5 * 16 and info["n_tris"] == 80 and info["n_bvh_nodes"] > 20


@pytest.mark.skipif(not ref_pbrt_available(), reason="oracle/_ref/libptmi_ref_pbrt.so not built (needs /root/reference)")
def test_live_against_the_compiled_reference(tmp_path):
    for f in sorted(glob.glob(os.path.join(PBRT, "*.pbrt"))):
        want, got = ref_pbrt_load(f), product_load(f)
        assert (want is None) == (got is None), os.path.basename(f)
        if want is not None:
            same(got, want, os.path.basename(f))
    n_loaded = 0
    for seed in range(300):
        f = tmp_path / f"fuzz{seed}.pbrt"
        f.write_text(random_scene(seed))
        want, got = ref_pbrt_load(str(f)), product_load(str(f))
        assert (want is None) == (got is None), seed
        if want is not None:
            same(got, want, f"fuzz seed {seed}")
            n_loaded += 1
    assert n_loaded > 200
    n_loaded = 0
    for seed in range(200):                                   # the same scenes plus "plymesh" shapes over random .ply files
        f = tmp_path / f"plyfuzz{seed}.pbrt"
        f.write_text(random_scene_with_ply(seed, str(tmp_path)))
        want, got = ref_pbrt_load(str(f)), product_load(str(f))
        assert (want is None) == (got is None), seed
        if want is not None:
            same(got, want, f"ply fuzz seed {seed}")
            n_loaded += 1
    assert n_loaded > 120
    n_loaded = n_rejected = 0
    for seed in range(150):                                   # the same scenes plus materials with "spectrum" parameters from .spd files
        f = tmp_path / f"spdfuzz{seed}.pbrt"
        f.write_text(random_scene_with_spd(seed, str(tmp_path)))
        want, got = ref_pbrt_load(str(f)), product_load(str(f))
        assert (want is None) == (got is None), seed
        if want is not None:
            same(got, want, f"spd fuzz seed {seed}")
            n_loaded += 1
        else:
            n_rejected += 1
    assert n_loaded > 45 and n_rejected > 15


@pytest.mark.skipif(not ref_pbrt_available(), reason="oracle/_ref/libptmi_ref_pbrt.so not built (needs /root/reference)")
def test_oversized_scene_becomes_the_bounding_box_proxy(tmp_path):
    """more than 2,000,000 triangles: a 12-triangle box over the scene bounds (utils/pbrt_loader.h:205-270)"""
    f = tmp_path / "big.pbrt"
    with open(f, "w") as out:
        out.write('WorldBegin\nTranslate 1 2 3\nRotate 30 0 1 0\nObjectBegin "m"\nShape "trianglemesh" "point P" [0 0 0  2 0 0.5  0 3 1  -1 -1 -2] "integer indices" [\n')
        out.write("0 1 2 1 2 3\n" * 500001)
        out.write(']\nObjectEnd\nObjectInstance "m"\nScale 2 2 2\nObjectInstance "m"\nWorldEnd\n')
    want, got = ref_pbrt_load(str(f)), product_load(str(f))
    assert want is not None and len(want["type"]) == 12
    same(got, want, "proxy")
    assert np.allclose(got["bsdf"], [[0.8, 0.2, 0.2]] * 12)
