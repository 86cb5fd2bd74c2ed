"""Second, independent restatement of the reference shader in NumPy float32.

TEST INFRASTRUCTURE ONLY.  Written straight from the WGSL (entry.wgsl, julia.wgsl,
gen_julia.wgsl, kifs.wgsl, quaternions.wgsl), vectorised over the frame with masks, using
NumPy's own sqrt / log / log2 / power / arccos / sin / cos, true divisions everywhere and NO
fused multiply-adds -- i.e. a different but equally legal evaluation of the same shader (WGSL
leaves contraction and builtin precision to the implementation).  In particular it keeps the
LITERAL form of everything the C oracle's arithmetic contract rewrote: `dot(n, p) / length(n)`
in the mirror, and in quat_pow (quaternions.wgsl:57-63) two separate lengths, `real / norm`,
`normalize(ijk)` and `pow(norm, x)`, with gen_julia.wgsl:16's `pow(q_sq_norm, power - 1)` taken
on its own -- no shared log2, no reciprocals.
Its job is to catch transcription errors in the C oracle (a wrong sign, a swapped
column): the two must agree on almost every pixel, differing only where a 1-ulp change
flips a `distance < epsilon` decision.  It is not bit-compatible with the C oracle or
the HIP kernel and is never used as the parity reference.
"""
import numpy as np

F = np.float32


def _f(x):
    return np.asarray(x, dtype=F)


def _dot(a, b):
    acc = a[0] * b[0]
    for i in range(1, len(a)):
        acc = acc + a[i] * b[i]
    return acc


def _length(v):
    return np.sqrt(_dot(v, v))


def _normalize(v):
    l = _length(v)
    with np.errstate(divide="ignore", invalid="ignore"):
        return [c / l for c in v]


class Scene:
    """Uniform values as python/NumPy scalars (taken from the oracle ctypes structs)."""

    def __init__(self, screen, camera, options, iters):
        self.width, self.height = int(screen.width), int(screen.height)
        self.h = F(screen.height)
        self.aspect = F(screen.aspect_ratio)
        self.origin = [F(camera.origin[i]) for i in range(3)]
        self.m = [[F(camera.matrix[c][r]) for r in range(3)] for c in range(3)]
        self.max_iterations = int(options.max_iterations)
        self.max_distance = F(options.max_distance)
        self.epsilon = F(options.epsilon)
        self.fractal = [F(options.fractal_color[i]) for i in range(3)]
        self.background = [F(options.background_color[i]) for i in range(3)]
        self.is_heatmap = bool(options.is_heatmap)
        self.group = int(options.fractal_group_id)
        self.primitive = int(options.primitive_id)
        self.power = F(options.power)
        self.c = [F(options.constant[i]) for i in range(4)]
        self.sdf_iters, self.normal_iters, self.fold_iters = (
            int(iters.sdf_iters), int(iters.normal_iters), int(iters.fold_iters))


# ---- quaternions.wgsl -----------------------------------------------------------------
def quat_sq(q):
    r, ijk = q[0], q[1:]
    return [r * r - _dot(ijk, ijk)] + [F(2.0) * r * c for c in ijk]


def quat_add(a, b):
    return [x + y for x, y in zip(a, b)]


def quat_pow(q, x):
    """quaternions.wgsl:57-63, literally: norm = length(q); phi = acos(real / norm);
    n = normalize(ijk); pow(norm, x) * (cos(x phi), n sin(x phi))."""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        norm = _length(q)                       # quat_norm2, :18-20
        phi = np.arccos(q[0] / norm).astype(F)
        n = _normalize(q[1:])
        pw = np.power(norm, x).astype(F)
        a = x * phi
        cs, sn = np.cos(a).astype(F), np.sin(a).astype(F)
        return [pw * cs] + [pw * (c * sn) for c in n]


# ---- gen_julia.wgsl ----------------------------------------------------------------------
def genjulia_sdf(s, p):  # gen_julia.wgsl:5-27
    norm = _length(p)
    outside = norm > F(2.0) + s.epsilon
    res = norm - F(2.0)
    idx = np.nonzero(~outside)[0]
    if idx.size:
        q = [p[0][idx], p[1][idx], p[2][idx], np.full(idx.size, 0.1, dtype=F)]
        qs = _dot(q, q)
        dqs = np.ones(idx.size, dtype=F)
        live = np.ones(idx.size, dtype=bool)
        power = s.power
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            for _ in range(s.sdf_iters):
                if not live.any():
                    break
                # :16  dq_sq_norm *= power * power * pow(q_sq_norm, power - 1)   (left to right)
                factor = power * power * np.power(qs, power - F(1.0)).astype(F)
                dqs = np.where(live, dqs * factor, dqs)
                nq = quat_add(quat_pow(q, power), s.c)
                q = [np.where(live, a, b) for a, b in zip(nq, q)]
                qs = np.where(live, _dot(q, q), qs)
                live = live & ~(qs > s.max_distance)
            val = F(0.25) * np.log(qs) * np.sqrt(qs / dqs)
        res = res.copy()
        res[idx] = val.astype(F)
    return res


def genjulia_normal(s, p):  # gen_julia.wgsl:30-55
    h, z = s.epsilon, F(0.0)
    n = p[0].size
    w = np.full(n, 0.1, dtype=F)
    qs = []
    for ax in range(3):
        off = [h if i == ax else z for i in range(3)]
        qs.append([pc + o for pc, o in zip(p, off)] + [w])
        qs.append([pc - o for pc, o in zip(p, off)] + [w])
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        for _ in range(s.normal_iters):
            qs = [quat_add(quat_pow(q, s.power), s.c) for q in qs]
        l = [np.log2(_length(q)).astype(F) for q in qs]
        return _normalize([l[0] - l[1], l[2] - l[3], l[4] - l[5]])


# ---- julia.wgsl -----------------------------------------------------------------------
def julia_sdf(s, p):
    norm = _length(p)
    outside = norm > F(2.0) + s.epsilon
    res = norm - F(2.0)
    idx = np.nonzero(~outside)[0]
    if idx.size:
        q = [p[0][idx], p[1][idx], p[2][idx], np.full(idx.size, 0.1, dtype=F)]
        qs = _dot(q, q)
        dqs = np.ones(idx.size, dtype=F)
        live = np.ones(idx.size, dtype=bool)
        with np.errstate(over="ignore", invalid="ignore"):
            for _ in range(s.sdf_iters):
                if not live.any():
                    break
                dqs = np.where(live, dqs * (F(4.0) * qs), dqs)
                nq = quat_add(quat_sq(q), s.c)
                q = [np.where(live, a, b) for a, b in zip(nq, q)]
                qs = np.where(live, _dot(q, q), qs)
                live = live & ~(qs > s.max_distance)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            val = F(0.25) * np.log(qs) * np.sqrt(qs / dqs)
        res = res.copy()
        res[idx] = val.astype(F)
    return res


def julia_normal(s, p):
    n = p[0].size
    q = [p[0].copy(), p[1].copy(), p[2].copy(), np.full(n, 0.1, dtype=F)]
    # J[col][row]
    J = [[np.full(n, 1.0 if r == c else 0.0, dtype=F) for r in range(4)] for c in range(4)]
    live = np.ones(n, dtype=bool)
    zero = np.zeros(n, dtype=F)
    with np.errstate(over="ignore", invalid="ignore"):
        for _ in range(s.normal_iters):
            if not live.any():
                break
            x, y, z, w = q
            # mat4x4 constructor takes COLUMNS: A[col][row]
            A = [[x, -y, -z, -w], [y, x, zero, zero], [z, zero, x, zero], [w, zero, zero, x]]
            newJ = []
            for c in range(4):  # (A*J)[c] = sum_k A[k] * J[c][k]
                col = []
                for r in range(4):
                    acc = A[0][r] * J[c][0]
                    for k in range(1, 4):
                        acc = acc + A[k][r] * J[c][k]
                    col.append(acc)
                newJ.append(col)
            J = [[np.where(live, a, b) for a, b in zip(nc, oc)] for nc, oc in zip(newJ, J)]
            nq = quat_add(quat_sq(q), s.c)
            q = [np.where(live, a, b) for a, b in zip(nq, q)]
            live = live & ~(_dot(q, q) > s.max_distance)
    g = []
    for r in range(3):
        acc = J[0][r] * q[0]
        for k in range(1, 4):
            acc = acc + J[k][r] * q[k]
        g.append(acc)
    return _normalize(g)


# ---- kifs.wgsl ------------------------------------------------------------------------
def _plane_mirror(normal, p):
    nl = _length([F(c) for c in normal])
    sdf = _dot([F(c) for c in normal], p) / nl
    d = np.minimum(sdf, F(0.0))
    nn = [F(c) / nl for c in normal]
    return [pc - F(2.0) * d * nc for pc, nc in zip(p, nn)]


def _fold(p):
    normal = (1.0, 1.0, 0.0)
    for _ in range(3):
        p = _plane_mirror(normal, p)
        normal = (normal[2], normal[0], normal[1])  # .zxy
    return p


def kifs_sdf(s, p):
    prim = s.primitive
    if prim == 0:
        return _length(p) - F(1.0)
    if prim == 1:
        d = [np.abs(_length(p[:2])) - F(1.0), np.abs(p[2]) - F(2.0)]
        return np.minimum(np.maximum(d[0], d[1]), F(0.0)) + _length([np.maximum(c, F(0.0)) for c in d])
    if prim == 2:
        q = [np.abs(c) - F(1.0) for c in p]
        return _length([np.maximum(c, F(0.0)) for c in q]) + \
            np.minimum(np.maximum(q[0], np.maximum(q[1], q[2])), F(0.0))
    if prim == 3:
        q = [_length(p[:2]) - F(1.0), p[2]]
        return _length(q) - F(0.3)
    if prim == 4:
        scale = np.ones(p[0].size, dtype=F)
        pos = [c.copy() for c in p]
        r = _length(pos)
        for _ in range(s.fold_iters):
            live = r < s.max_distance
            if not live.any():
                break
            np_ = _fold(pos)
            np_ = [F(2.0) * c - F(1.0) for c in np_]
            pos = [np.where(live, a, b) for a, b in zip(np_, pos)]
            scale = np.where(live, scale * F(2.0), scale)
            r = np.where(live, _length(pos), r)
        return (r - F(2.0)) / scale
    if prim == 5:
        return bunny_sdf(p)
    return np.ones(p[0].size, dtype=F)


_BUNNY = None


def _bunny_tables():
    """The network's weights: DATA, read from the generated table the C oracle includes
    (oracle/kifs_oracle_bunny.inc <- tools/gen_bunny_tables.py <- the literals of kifs.wgsl:90-136).
    Only the numbers are shared; how they are combined below is this file's own reading."""
    global _BUNNY
    if _BUNNY is None:
        import re
        from pathlib import Path
        text = (Path(__file__).resolve().parent / "kifs_oracle_bunny.inc").read_text()
        tabs = {}
        for name, body in re.findall(r"KOR_BUNNY_(\w+)(?:\[\d+\])+\s*=\s*\{(.*?)\};", text, flags=re.S):
            vals = [float(v) for v in re.findall(r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?(?=f)", body)]
            tabs[name] = np.array(vals, dtype=F)
        _BUNNY = {"L0": tabs["L0"].reshape(4, 4, 4), "L1": tabs["L1"].reshape(4, 4, 4, 4),
                  "L2": tabs["L2"].reshape(4, 4, 4, 4), "B1": tabs["B1"].reshape(4, 4),
                  "B2": tabs["B2"].reshape(4, 4), "OUT": tabs["OUT"].reshape(4, 4)}
    return _BUNNY


def _mat_vec(m, v):
    """Column-major mat4x4f (m[col][row]) times vec4 v: sum over columns of column * v[col], first
    column first (no fma)."""
    out = []
    for r in range(4):
        acc = m[0][r] * v[0]
        for c in range(1, 4):
            acc = acc + m[c][r] * v[c]
        out.append(acc)
    return out


def bunny_sdf(p):  # kifs.wgsl:84-137
    T = _bunny_tables()
    d2 = _dot(p, p)
    far = d2 > F(1.0)
    res = (_length(p) - F(0.8)).astype(F)
    idx = np.nonzero(~far)[0]
    if idx.size:
        # q = vec4(position.xzy * (-1, 1, -1), 1)
        q = [p[0][idx] * F(-1.0), p[2][idx] * F(1.0), p[1][idx] * F(-1.0), np.ones(idx.size, dtype=F)]
        f0 = [[np.sin(c).astype(F) for c in _mat_vec(T["L0"][u], q)] for u in range(4)]

        def layer(W, bias, prev, scale):
            out = []
            for u in range(4):
                acc = _mat_vec(W[u][0], prev[0])
                for k in range(1, 4):
                    acc = [a + b for a, b in zip(acc, _mat_vec(W[u][k], prev[k]))]
                acc = [a + F(bias[u][r]) for r, a in enumerate(acc)]
                sn = [np.sin(a).astype(F) for a in acc]
                if scale is not None:
                    sn = [v / scale for v in sn]            # `sin(...) / 1.4 + f1k`: a true division of the sine
                out.append([v + prev[u][r] for r, v in enumerate(sn)])
            return out
        f1 = layer(T["L1"], T["B1"], f0, None)
        f2 = layer(T["L2"], T["B2"], f1, F(1.4))
        total = _dot(f2[0], [F(x) for x in T["OUT"][0]])
        for u in range(1, 4):
            total = total + _dot(f2[u], [F(x) for x in T["OUT"][u]])
        res = res.copy()
        res[idx] = (total - F(0.16)).astype(F)
    return res


def kifs_normal(s, p):
    h = s.epsilon
    z = F(0.0)
    d = []
    for ax in range(3):
        off = [h if i == ax else z for i in range(3)]
        a = kifs_sdf(s, [pc + o for pc, o in zip(p, off)])
        b = kifs_sdf(s, [pc - o for pc, o in zip(p, off)])
        d.append(a - b)
    return _normalize(d)


def scene_sdf(s, p):
    return julia_sdf(s, p) if s.group == 1 else genjulia_sdf(s, p) if s.group == 2 else kifs_sdf(s, p)


def scene_normal(s, p):
    return julia_normal(s, p) if s.group == 1 else genjulia_normal(s, p) if s.group == 2 else kifs_normal(s, p)


# ---- entry.wgsl -----------------------------------------------------------------------
def render_linear(screen, camera, options, iters, y0=0, y1=None):
    """Linear RGBA f32 (H, W, 4) and the loop counter i (H, W).  Every pipeline: Julia, generalised
    Julia, the KIFS primitives, Sierpinski and the bunny."""
    s = Scene(screen, camera, options, iters)
    assert s.group in (0, 1, 2)
    W = s.width
    y1 = s.height if y1 is None else y1
    H = y1 - y0  # rows of the band [y0, y1); pixel coordinates stay global
    ys, xs = np.mgrid[y0:y1, 0:W]
    px = (xs.ravel().astype(F) + F(0.5))
    py = (ys.ravel().astype(F) + F(0.5))
    uvx = F(2.0) * px / s.h - s.aspect
    uvy = F(2.0) * py / s.h - F(1.0)
    d = [uvx * s.m[1][k] - uvy * s.m[2][k] - s.m[0][k] for k in range(3)]
    dirv = _normalize(d)
    n = W * H
    t = np.zeros(n, dtype=F)
    pos = [np.full(n, s.origin[k], dtype=F) for k in range(3)]
    i = np.zeros(n, dtype=np.int32)
    hit = np.zeros(n, dtype=bool)
    live = np.full(n, s.max_iterations > 0, dtype=bool) & (t < s.max_distance)
    while live.any():
        idx = np.nonzero(live)[0]
        p = [c[idx] for c in pos]
        dist = scene_sdf(s, p)
        with np.errstate(invalid="ignore"):
            h = dist < s.epsilon
        hit[idx[h]] = True
        go = idx[~h]
        t[go] = t[go] + dist[~h]
        for k in range(3):
            pos[k][go] = s.origin[k] + t[go] * dirv[k][go]
        i[go] += 1
        live[idx[h]] = False
        with np.errstate(invalid="ignore"):
            live[go] = (i[go] < s.max_iterations) & (t[go] < s.max_distance)
    rgb = [np.full(n, s.background[k], dtype=F) for k in range(3)]
    hidx = np.nonzero(hit)[0]
    if hidx.size:
        nrm = scene_normal(s, [c[hidx] for c in pos])
        with np.errstate(invalid="ignore"):
            ndl = nrm[0] * F(1.0) + nrm[1] * F(1.0) + nrm[2] * F(1.0)
            diffuse = F(0.1) + F(0.9) * np.minimum(np.maximum(ndl, F(0.0)), F(1.0))
        for k in range(3):
            rgb[k][hidx] = diffuse * s.fractal[k]
    if s.is_heatmap:
        with np.errstate(invalid="ignore", divide="ignore"):
            f = i.astype(F) / F(s.max_iterations)
        rgb = [f * s.fractal[k] for k in range(3)]
    out = np.stack(rgb + [np.ones(n, dtype=F)], axis=-1).reshape(H, W, 4)
    return out, i.reshape(H, W), hit.reshape(H, W)


def srgb_encode_ideal(x):
    """Round-to-nearest sRGB UNORM8 of linear x, evaluated in float64."""
    x = np.nan_to_num(np.asarray(x, dtype=np.float64), nan=0.0)
    x = np.clip(x, 0.0, 1.0)
    v = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1.0 / 2.4) - 0.055)
    return np.floor(v * 255.0 + 0.5).astype(np.uint8)


def render(screen, camera, options, iters, encode=1, y0=0, y1=None):
    lin, i, hit = render_linear(screen, camera, options, iters, y0, y1)
    if encode == 1:
        rgb = srgb_encode_ideal(lin[..., :3])
    else:
        x = np.clip(np.nan_to_num(lin[..., :3].astype(np.float64), nan=0.0), 0, 1)
        rgb = np.floor(x * 255.0 + 0.5).astype(np.uint8)
    a = np.full(lin.shape[:2] + (1,), 255, dtype=np.uint8)
    return np.concatenate([rgb, a], axis=-1), i, hit
