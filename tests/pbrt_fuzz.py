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
        elif r < 0.75: out.append(f'AreaLightSource "diffuse" "rgb L" [{_vec(rng).replace("-", "")}]')
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
