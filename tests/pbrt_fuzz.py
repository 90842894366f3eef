"""Random PBRT scenes for the differential test of the product's loader against the compiled reference importer."""
import numpy as np


def _num(rng):
    style = rng.integers(6)
    v = rng.normal(0, 2.0)
    if style == 0: return f"{v:.3f}"
    if style == 1: return f"{v:.6e}"
    if style == 2: return str(int(rng.integers(-4, 5)))
    if style == 3: return f"{v:.1f}"
    if style == 4: return ("-0.0" if rng.random() < 0.5 else "0")
    return f"{abs(v):.2f}".lstrip("0") or "0"


def _vec(rng, n=3):
    return " ".join(_num(rng) for _ in range(n))


def _transform(rng):
    k = rng.integers(7)
    if k == 0: return f"Translate {_vec(rng)}"
    if k == 1: return f"Scale {' '.join(f'{rng.uniform(0.25, 3.0) * rng.choice([1, 1, 1, -1]):.3f}' for _ in range(3))}"
    if k == 2: return f"Rotate {rng.uniform(-360, 360):.4f} {' '.join(f'{rng.normal():.3f}' for _ in range(3))}"
    if k == 3:
        m = np.eye(4); m[:3, :3] += rng.normal(0, 0.3, (3, 3)); m[3, :3] = rng.normal(0, 1, 3)
        return "ConcatTransform [" + " ".join(f"{x:.4f}" for x in m.reshape(-1)) + "]"
    if k == 4:
        m = np.eye(4); m[:3, :3] = np.diag(rng.uniform(0.5, 2, 3)); m[3, :3] = rng.normal(0, 1, 3)
        return "Transform [" + " ".join(f"{x:.4f}" for x in m.reshape(-1)) + "]"
    if k == 5: return "Identity" if rng.random() < 0.3 else f"Translate {_vec(rng)}"
    return "ReverseOrientation"


def _material(rng):
    k = rng.integers(9)
    c = lambda: " ".join(f"{rng.uniform(0, 1):.3f}" for _ in range(3))
    return [f'Material "matte" "rgb Kd" [{c()}]', f'Material "plastic" "color Kd" [{c()}] "rgb Ks" [{c()}] "float roughness" [{rng.uniform(0, 1):.3f}]',
            f'Material "mirror" "rgb Kr" [{c()}]', f'Material "glass" "rgb Kt" [{c()}]', f'Material "metal" "rgb eta" [{c()}] "rgb k" [{c()}]',
            f'Material "disney" "rgb color" [{c()}] "float metallic" [{rng.uniform(0, 1):.3f}]', f'Material "uber" "rgb Kd" [{c()}] "rgb Ks" [{c()}]',
            f'Material "substrate" "rgb Kd" [{c()}]', 'Material "matte"'][k]


def _mesh(rng):
    nv = int(rng.integers(3, 9)); nf = int(rng.integers(1, 6))
    P = " ".join(_vec(rng) for _ in range(nv))
    idx = " ".join(str(int(i)) for i in rng.integers(0, nv, nf * 3))
    s = f'Shape "trianglemesh" "integer indices" [{idx}] "point P" [{P}]'
    if rng.random() < 0.5:
        s += ' "normal N" [' + " ".join(" ".join(f"{x:.3f}" for x in rng.normal(0, 1, 3) + 0.01) for _ in range(nv if rng.random() < 0.8 else nv - 1)) + "]"
    return s


_PLY_TYPES = {"char": ("b", -128, 127), "uchar": ("B", 0, 255), "short": ("h", -32768, 32767), "ushort": ("H", 0, 65535),
              "int": ("i", -2 ** 31, 2 ** 31 - 1), "uint": ("I", 0, 2 ** 32 - 1), "float": ("f", None, None), "double": ("d", None, None),
              "int8": ("b", -128, 127), "uint8": ("B", 0, 255), "int16": ("h", -32768, 32767), "uint16": ("H", 0, 65535),
              "int32": ("i", -2 ** 31, 2 ** 31 - 1), "uint32": ("I", 0, 2 ** 32 - 1), "float32": ("f", None, None), "float64": ("d", None, None)}


def random_ply(rng, path, broken=0):
    """A random PLY file for a "plymesh" shape: ASCII or either binary byte order, properties in random order with extras
    between them, optional normals / texture coordinates, an extra element before, between or after vertex and face, comments.
    broken: 1 = a quad among the faces, 2 = truncated data, 3 = header without format line, 4 = unknown property type,
    5 = a vertex coordinate missing, 6 = out-of-range ASCII value (uchar 300)."""
    import struct
    fmt = ["ascii", "binary_little_endian", "binary_big_endian"][int(rng.integers(3))]
    if broken == 6: fmt = "ascii"
    nv = int(rng.integers(3, 12)); nf = int(rng.integers(1, 8))
    ftype = lambda: ["float", "float32", "double", "float64"][int(rng.integers(4))]
    vprops = [("x", ftype()), ("y", ftype()), ("z", ftype())]
    if broken == 5: vprops.pop(int(rng.integers(3)))
    if rng.random() < 0.5: vprops += [("nx", ftype()), ("ny", ftype()), ("nz", ftype())]
    elif rng.random() < 0.2: vprops += [("nx", ftype()), ("nz", ftype())]                      # incomplete normals: ignored
    if rng.random() < 0.4: vprops += [(["u", "s", "texture_u"][int(rng.integers(3))], "float"), (["v", "t", "texture_v"][int(rng.integers(3))], "float")]
    for _ in range(int(rng.integers(0, 3))): vprops.append((f"extra{len(vprops)}", list(_PLY_TYPES)[int(rng.integers(16))]))
    if rng.random() < 0.3: vprops.append(("tags", ("list", "uchar", list(_PLY_TYPES)[int(rng.integers(16))])))
    order = rng.permutation(len(vprops)); vprops = [vprops[i] for i in order]
    ctype = ["uchar", "uint8", "int", "ushort", "char"][int(rng.integers(5))]
    itype = ["int", "uint", "int32", "ushort", "uchar", "float", "short"][int(rng.integers(7))]
    fprops = [(["vertex_indices", "vertex_index"][int(rng.integers(2))], ("list", ctype, itype))]
    if rng.random() < 0.3: fprops.insert(int(rng.integers(2)), ("flags", "uchar"))
    if rng.random() < 0.2: fprops.append(("texcoord", ("list", "uchar", "float")))
    elements = [("vertex", nv, vprops), ("face", nf, fprops)]
    if rng.random() < 0.3: elements.reverse()
    if rng.random() < 0.4: elements.insert(int(rng.integers(3)), ("edge", int(rng.integers(0, 4)), [("a", "int"), ("b", "int"), ("w", "float")]))
    nl = "\r\n" if rng.random() < 0.15 else "\n"
    head = ["ply"]
    if broken != 3: head.append(f"format {fmt} 1.0")
    if rng.random() < 0.5: head.append("comment generated for the differential test")
    for name, n, props in elements:
        head.append(f"element {name} {n}")
        for pn, pt in props:
            if isinstance(pt, tuple): head.append(f"property list {pt[1]} {pt[2]} {pn}")
            else: head.append(f"property {'floot' if broken == 4 and pn == 'y' else pt} {pn}")
        if rng.random() < 0.2: head.append("obj_info something")
    head.append("end_header")
    data = bytearray((nl.join(head) + nl).encode())
    en = "<" if fmt == "binary_little_endian" else ">"
    words = []

    def put(t, v):
        code = _PLY_TYPES[t][0]
        if fmt == "ascii":
            words.append(repr(float(v)) if code in "fd" and rng.random() < 0.7 else (f"{v:.4g}" if code in "fd" else str(int(v))))
        else:
            data.extend(struct.pack(en + code, float(v) if code in "fd" else int(v)))

    def scalar(t):
        code, lo, hi = _PLY_TYPES[t]
        return float(np.float32(rng.normal(0, 2))) if code in "fd" else int(rng.integers(max(lo, -1000), min(hi, 1000) + 1))

    quad_at = int(rng.integers(nf)) if broken == 1 else -1
    for name, n, props in elements:
        for j in range(n):
            for pn, pt in props:
                if isinstance(pt, tuple):
                    if pn.startswith("vertex_ind"):
                        k = 4 if j == quad_at else 3
                        put(pt[1], k)
                        for _ in range(k): put(pt[2], int(rng.integers(nv)))
                    else:
                        k = int(rng.integers(0, 4)); put(pt[1], k)
                        for _ in range(k): put(pt[2], scalar(pt[2]))
                else:
                    put(pt, 300 if broken == 6 and pt in ("uchar", "uint8") else scalar(pt))
            if fmt == "ascii": words.append("\n")
    if fmt == "ascii":
        data.extend(" ".join(words).encode())
    if broken == 2: data = data[: max(len(data) - int(rng.integers(1, 9)), len(nl.join(head)) + 1)]
    with open(path, "wb") as f: f.write(bytes(data))


def random_scene_with_ply(seed, directory):
    """random_scene plus "plymesh" shapes; writes the .ply files next to the scene and returns (scene text, expect_failure_hint)"""
    import os
    rng = np.random.default_rng(10_000 + seed)
    text = random_scene(seed).rstrip("\n").split("\n")
    assert text[-1] == "WorldEnd"
    body = text[:-1]
    n = int(rng.integers(1, 4))
    for k in range(n):
        broken = int(rng.integers(1, 7)) if rng.random() < 0.15 else 0
        name = f"mesh{seed}_{k}.ply"
        random_ply(rng, os.path.join(directory, name), broken)
        line = f'Shape "plymesh" "string filename" "{name}"'
        if rng.random() < 0.3: line = "AttributeBegin\n" + _transform(rng) + "\n" + _material(rng) + "\n" + line + "\nAttributeEnd"
        body.append(line)
    return "\n".join(body + ["WorldEnd"]) + "\n"


def random_scene_with_spd(seed, directory):
    """random_scene plus materials whose "spectrum" parameters name .spd files (written next to the scene): metals with eta / k as
    pairs or as three numbers, a matte Kd that must be three numbers, comments, quoted numbers, words that are no numbers, an odd
    count, a file that is not there, an absolute path"""
    import os
    rng = np.random.default_rng(20_000 + seed)
    text = random_scene(seed).rstrip("\n").split("\n")
    assert text[-1] == "WorldEnd"
    body = text[:-1]

    def spd(tag):
        kind = rng.choice(["pairs", "pairs", "pairs", "three", "odd", "word", "quoted", "empty", "missing"], p=[0.3, 0.15, 0.15, 0.15, 0.05, 0.05, 0.05, 0.05, 0.05])
        name = f"s{seed}_{tag}.spd"
        n = {"three": 3, "odd": int(rng.choice([1, 5, 7])), "empty": 0}.get(kind, 2 * int(rng.integers(1, 9)))
        vals = [f"{rng.uniform(300, 800):.4f}" if i % 2 == 0 else f"{rng.uniform(0.05, 5):.5f}" for i in range(n)]
        if kind == "word" and vals: vals[int(rng.integers(len(vals)))] = "nm"
        if kind == "quoted" and vals: i = int(rng.integers(len(vals))); vals[i] = '"' + vals[i] + '"'
        if kind != "missing":
            with open(os.path.join(directory, name), "w") as f:
                if rng.random() < 0.5: f.write("# measured data\n")
                f.write((" " if rng.random() < 0.5 else "\n").join(vals) + ("\n" if rng.random() < 0.7 else ""))
        return os.path.join(directory, name) if rng.random() < 0.2 else name
    tri = lambda z: f'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 {z}  1 0 {z}  0 1 {z}]'
    for k in range(int(rng.integers(1, 4))):
        r = rng.random()
        if r < 0.6: m = f'Material "metal" "spectrum eta" "{spd(f"{k}e")}" "spectrum k" "{spd(f"{k}k")}"'
        elif r < 0.8: m = f'Material "metal" "spectrum eta" "{spd(f"{k}e")}"'
        else: m = f'Material "matte" "spectrum Kd" "{spd(f"{k}d")}"'
        body += [m, tri(20 + k)]
    return "\n".join(body + ["WorldEnd"]) + "\n"


def random_scene(seed):
    rng = np.random.default_rng(seed)
    out = ["# fuzz scene %d" % seed, f"LookAt {_vec(rng, 3)} 0 0 0 0 1 0", 'Camera "perspective" "float fov" [45]', "WorldBegin"]
    names = []
    depth = 0
    in_object = False
    for _ in range(int(rng.integers(8, 40))):
        r = rng.random()
        if r < 0.25: out.append(_mesh(rng))
        elif r < 0.45: out.append(_transform(rng))
        elif r < 0.55: out.append(_material(rng))
        elif r < 0.63: out.append("AttributeBegin"); depth += 1
        elif r < 0.70 and depth > 0: out.append("AttributeEnd"); depth -= 1
        elif r < 0.73: out.append(f'AreaLightSource "diffuse" "rgb L" [{_vec(rng).replace("-", "")}]')
        elif r < 0.75: out.append(f'AreaLightSource "diffuse" "blackbody L" [{rng.uniform(800, 15000):.2f} {rng.uniform(0.1, 50):.3f}]')
        elif r < 0.82 and not in_object:
            n = f"obj{len(names)}"; names.append(n); out.append(f'ObjectBegin "{n}"'); in_object = True; obj_depth = depth
        elif r < 0.88 and in_object and depth == obj_depth: out.append("ObjectEnd"); in_object = False
        elif r < 0.96 and names:
            # an object may instance earlier, closed objects only (self-instancing sends the reference into unbounded recursion)
            closed = names[:-1] if in_object else names
            if closed: out.append(f'ObjectInstance "{closed[int(rng.integers(len(closed)))]}"')
        elif r < 0.98: out.append('Shape "sphere" "float radius" [1.5]')
        else: out.append("TransformBegin\n" + _transform(rng) + "\n" + _mesh(rng) + "\nTransformEnd")
    if in_object:
        while depth > obj_depth: out.append("AttributeEnd"); depth -= 1
        out.append("ObjectEnd")
    while depth > 0: out.append("AttributeEnd"); depth -= 1
    out.append("WorldEnd")
    return "\n".join(out) + "\n"
