"""Procedural scenes for the stress configurations (BASELINE.json configs[4]).

tessellated_cornell(cells_u, cells_v): every quad of a quad-only scene (cbox_quads.obj) is cut into a
cells_u x cells_v grid, two triangles per cell, and every grid vertex is pushed along the quad's normal by
1e-3 * (hash32(vertex_id, seed) / 2^32 - 0.5) so that bounding boxes are not degenerate.  256 x 128 cells on the
16 quads give 1,048,576 triangles.  Materials are inherited; each triangle carries its geometric normal.
Returns arrays in the layout ptmi_load_scene_arrays / po_scene_from_arrays take.
"""
import numpy as np


def hash32(x, seed):
    """lowbias32-style integer hash, vectorised (uint32 in, uint32 out)."""
    x = (np.asarray(x, np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def tessellated_cornell(prims, cells_u=256, cells_v=128, seed=1, amplitude=1e-3):
    """prims: dict(type, verts, normal, bsdf, Le) of a quad-only scene (e.g. ptmi.HostScene.load(cbox_quads).prims())."""
    F = np.float32
    types, verts = prims["type"], prims["verts"].astype(F)
    assert (types == 1).all(), "expects a quad-only scene"
    nq = len(types)
    s = (np.arange(cells_u + 1, dtype=F) / F(cells_u))[None, :, None]      # along v00 -> v10
    t = (np.arange(cells_v + 1, dtype=F) / F(cells_v))[:, None, None]      # along v00 -> v01
    out_v, out_n, out_b, out_e = [], [], [], []
    for q in range(nq):
        v00, v10, v11, v01 = verts[q]
        # bilinear patch through the four corners
        P = ((F(1) - s) * (F(1) - t)) * v00 + (s * (F(1) - t)) * v10 + (s * t) * v11 + ((F(1) - s) * t) * v01
        n = np.cross(v10 - v00, v01 - v00).astype(np.float64); n = (n / np.linalg.norm(n)).astype(F)
        vid = (q * (cells_u + 1) * (cells_v + 1) + np.arange((cells_u + 1) * (cells_v + 1))).reshape(cells_v + 1, cells_u + 1)
        disp = (hash32(vid, seed).astype(np.float64) / 2.0 ** 32 - 0.5) * amplitude
        P = (P + disp[..., None].astype(F) * n).astype(F)
        p00, p10, p11, p01 = P[:-1, :-1], P[:-1, 1:], P[1:, 1:], P[1:, :-1]
        tri = np.stack([np.stack([p00, p10, p11], -2), np.stack([p00, p11, p01], -2)], 2).reshape(-1, 3, 3)
        out_v.append(tri)
        gn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(np.float64)
        gn /= np.maximum(np.linalg.norm(gn, axis=1, keepdims=True), 1e-30)
        out_n.append(gn.astype(F))
        out_b.append(np.repeat(prims["bsdf"][q][None], len(tri), 0)); out_e.append(np.repeat(prims["Le"][q][None], len(tri), 0))
    tri = np.concatenate(out_v)
    v4 = np.zeros((len(tri), 4, 3), F); v4[:, :3] = tri
    return dict(type=np.zeros(len(tri), np.int32), verts=v4, normal=np.concatenate(out_n).astype(F),
                bsdf=np.concatenate(out_b).astype(F), Le=np.concatenate(out_e).astype(F))
