"""Synthetic workloads: library-independent PRNG, the uniform
clouds of configs C2-C5, and a restatement of the reference's pillar-map generator
(config C1's input).

Everything here is deterministic numpy on the host so that the build container and
the GPU box regenerate identical inputs; nothing depends on libstdc++ or torch RNGs.
"""
from __future__ import annotations

import math

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n outputs of the splitmix64 stream for `seed`, starting at element `offset`.

    Counter-based (element i depends only on seed and i), so any slice of a huge
    stream can be produced without generating what precedes it.
    """
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01_f32(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """fp32 uniforms in [0,1): top 24 bits of splitmix64, exact in fp32."""
    return ((splitmix64(seed, n, offset) >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def uniform_rows_f64(seed: int, n: int, dim: int, lo: float, hi: float) -> np.ndarray:
    """(n, dim) doubles uniform in [lo, hi) carrying 48 random bits each: almost none of them is an fp32 value."""
    a = uniform01_f32(seed, n * dim).astype(np.float64)
    b = uniform01_f32(seed + 7919, n * dim).astype(np.float64)
    return (lo + (a + b * 2.0 ** -24) * (hi - lo)).reshape(n, dim)


def uniform_points(seed: int, n: int, lo: float, hi: float, offset: int = 0) -> np.ndarray:
    """(n,3) fp32 points uniform in [lo,hi)^3; element k of the stream feeds coordinate k%3 of point k//3."""
    u = uniform01_f32(seed, 3 * n, 3 * offset).reshape(n, 3)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)


def uniform_points_chunked(seed: int, n: int, lo: float, hi: float, chunk: int = 1 << 22):
    """Generator of consecutive (offset, block) pieces of uniform_points(seed, n, lo, hi)."""
    off = 0
    while off < n:
        m = min(chunk, n - off)
        yield off, uniform_points(seed, m, lo, hi, offset=off)
        off += m


def shuffled_order(seed: int, n: int) -> np.ndarray:
    """A deterministic permutation of range(n) (argsort of a splitmix64 stream)."""
    return np.argsort(splitmix64(seed, n), kind="stable").astype(np.int64)


# ----------------------------------------------------------------------------------------
# Reference pillar map (config C1).  Restates Planner/src/map_generator.cpp:16-125 with the
# libstdc++ pieces it relies on spelled out:
#   std::default_random_engine == minstd_rand0: x <- 16807 x mod (2^31 - 1), min 1
#   std::uniform_real_distribution<double>(a,b): a + (b-a) * generate_canonical<double,53>
#   generate_canonical with a 2147483646-value engine draws 2 numbers per double.
# Parity: UNPINNED (the reference holds no expected map; only the survey's point count).
# ----------------------------------------------------------------------------------------
class _MinStd0:
    M = 2147483647

    def __init__(self, seed: int):
        s = seed % self.M
        self.x = s if s != 0 else 1

    def __call__(self) -> int:
        self.x = (self.x * 16807) % self.M
        return self.x

    def canonical(self) -> float:
        r = 2147483646.0
        s = float(self() - 1)
        s += float(self() - 1) * r
        ret = s / (r * r)
        if ret >= 1.0:
            ret = math.nextafter(1.0, 0.0)
        return ret

    def uniform(self, a: float, b: float) -> float:
        return self.canonical() * (b - a) + a


def _cround(v: float) -> float:
    """C round(): half away from zero."""
    return math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5)


def pillar_map(x_init=-10.0, x_end=9.0, y_init=-10.0, y_end=9.0,
               x1=-15.0, x2=15.0, y1=-15.0, y2=15.0, h1=1.0, h2=8.0, w1=0.6, w2=2.0,
               res=0.1, num=120, seed=6) -> np.ndarray:
    """(N,3) fp32 obstacle cloud of Planner/launch/clean_demo.launch (defaults = its constants).

    Follows map_generator::generate_map (map_generator.cpp:16-87) and
    emplace_rect_to_map (:97-125); points are narrowed to fp32 as pcl::PointXYZ does.
    """
    eng = _MinStd0(seed)
    cyl: list[tuple[float, float, float]] = []
    pts: list[tuple[float, float, float]] = []

    def emplace(xa, ya, xb, yb, h_):
        xa_c, ya_c = int(_cround(xa / res)), int(_cround(ya / res))
        xb_c, yb_c = int(_cround(xb / res)), int(_cround(yb / res))
        h_c = int(_cround(h_ / res))
        xi = 1 if xa_c < xb_c else -1
        yi = 1 if ya_c < yb_c else -1
        if xa_c == xb_c:
            y_i = ya_c
            while y_i != yb_c:
                for h_i in range(1, h_c):
                    pts.append((xa_c * res, y_i * res, h_i * res))
                y_i += yi
        elif ya_c == yb_c:
            x_i = xa_c
            while x_i != xb_c:
                for h_i in range(1, h_c):
                    pts.append((x_i * res, ya_c * res, h_i * res))
                x_i += xi
        else:
            x_i = xa_c
            while x_i != xb_c + xi:
                y_i = ya_c
                while y_i != yb_c + yi:
                    pts.append((x_i * res, y_i * res, h_c * res))
                    y_i += yi
                x_i += xi

    for _ in range(num):
        x = eng.uniform(x1, x2)
        y = eng.uniform(y1, y2)
        w = eng.uniform(w1, w2)
        h = eng.uniform(h1, h2)
        if ((x - x_init) ** 2 + (y - y_init) ** 2 < 2 + w * w
                or (x - x_end) ** 2 + (y - y_end) ** 2 < 2 + w * w):
            continue
        if any((cx - x) ** 2 + (cy - y) ** 2 < (cw + w) ** 2 for cx, cy, cw in cyl):
            continue
        cyl.append((x, y, w))
        h = _cround(h / res) * res
        cap = []
        delta = 90
        phi = delta // 2
        while phi < 360 + delta // 2:
            xa = _cround((x + w * math.cos(math.pi / 180 * phi)) / res) * res
            ya = _cround((y + w * math.sin(math.pi / 180 * phi)) / res) * res
            xb = _cround((x + w * math.cos(math.pi / 180 * (phi + delta))) / res) * res
            yb = _cround((y + w * math.sin(math.pi / 180 * (phi + delta))) / res) * res
            cap.append((xa, ya))
            emplace(xa, ya, xb, yb, h)
            phi += delta
        emplace(cap[0][0], cap[0][1], cap[2][0], cap[2][1], 0)
        emplace(cap[0][0], cap[0][1], cap[2][0], cap[2][1], h)
    return np.asarray(pts, dtype=np.float64).astype(np.float32).reshape(-1, 3)


def _emplace_np(xa, ya, xb, yb, h_, res):
    """emplace_rect_to_map (map_generator.cpp:97-125) as index arithmetic: the (x, y, z) lattice indices of one wall / cap, in the
    reference's loop order"""
    xa_c, ya_c = int(_cround(xa / res)), int(_cround(ya / res))
    xb_c, yb_c = int(_cround(xb / res)), int(_cround(yb / res))
    h_c = int(_cround(h_ / res))
    xi = 1 if xa_c < xb_c else -1
    yi = 1 if ya_c < yb_c else -1
    if xa_c == xb_c:
        ys = np.arange(ya_c, yb_c, yi, dtype=np.int64)
        hs = np.arange(1, h_c, dtype=np.int64)
        Y, H = np.meshgrid(ys, hs, indexing="ij")
        return np.stack([np.full(Y.size, xa_c, np.int64), Y.ravel(), H.ravel()], 1)
    if ya_c == yb_c:
        xs = np.arange(xa_c, xb_c, xi, dtype=np.int64)
        hs = np.arange(1, h_c, dtype=np.int64)
        X, H = np.meshgrid(xs, hs, indexing="ij")
        return np.stack([X.ravel(), np.full(X.size, ya_c, np.int64), H.ravel()], 1)
    xs = np.arange(xa_c, xb_c + xi, xi, dtype=np.int64)
    ys = np.arange(ya_c, yb_c + yi, yi, dtype=np.int64)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), np.full(X.size, h_c, np.int64)], 1)


def pillar_map_scaled(scale: float = 1.0, seed: int = 6, res: float = 0.1) -> np.ndarray:
    """The reference's pillar world (map_generator.cpp:16-125 with clean_demo.launch's constants) on a square `scale` times as wide:
    the same pillar density (120 attempts per 30 m x 30 m), widths, heights and lattice, so the cloud has the reference's surface
    structure at any size -- scale 1 reproduces pillar_map() point for point (182,332 points); scale 7.5 gives ~10 M points.
    SURVEY 8(d)'s "clustered variant" of the large configs."""
    half = 15.0 * scale
    num = int(round(120 * scale * scale))
    x_init, y_init, x_end, y_end = -10.0 * scale, -10.0 * scale, 9.0 * scale, 9.0 * scale
    eng = _MinStd0(seed)
    cx, cy, cw = np.zeros(num), np.zeros(num), np.zeros(num)
    k = 0
    parts = []
    for _ in range(num):
        x = eng.uniform(-half, half)
        y = eng.uniform(-half, half)
        w = eng.uniform(0.6, 2.0)
        h = eng.uniform(1.0, 8.0)
        if ((x - x_init) ** 2 + (y - y_init) ** 2 < 2 + w * w or (x - x_end) ** 2 + (y - y_end) ** 2 < 2 + w * w):
            continue
        if k and np.any((cx[:k] - x) ** 2 + (cy[:k] - y) ** 2 < (cw[:k] + w) ** 2):
            continue
        cx[k], cy[k], cw[k] = x, y, w
        k += 1
        h = _cround(h / res) * res
        cap = []
        for phi in (45, 135, 225, 315):
            xa = _cround((x + w * math.cos(math.pi / 180 * phi)) / res) * res
            ya = _cround((y + w * math.sin(math.pi / 180 * phi)) / res) * res
            xb = _cround((x + w * math.cos(math.pi / 180 * (phi + 90))) / res) * res
            yb = _cround((y + w * math.sin(math.pi / 180 * (phi + 90))) / res) * res
            cap.append((xa, ya))
            parts.append(_emplace_np(xa, ya, xb, yb, h, res))
        parts.append(_emplace_np(cap[0][0], cap[0][1], cap[2][0], cap[2][1], 0, res))
        parts.append(_emplace_np(cap[0][0], cap[0][1], cap[2][0], cap[2][1], h, res))
    idx = np.concatenate(parts) if parts else np.zeros((0, 3), np.int64)
    return (idx.astype(np.float64) * res).astype(np.float32)


def crop_ball(points: np.ndarray, center, radius: float) -> np.ndarray:
    d = points.astype(np.float64) - np.asarray(center, dtype=np.float64)
    return points[(d * d).sum(1) <= radius * radius]


def clustered_points(seed: int, n: int, lo: float, hi: float, res: float = 0.1) -> np.ndarray:
    """n fp32 points on a `res` grid, bunched on vertical pillar faces (map-generator-like
    occupancy: most cells empty, many exactly equidistant pairs)."""
    k = max(1, n // 2048)
    ctr = uniform_points(seed ^ 0x5EED, k, lo + 2.0, hi - 2.0)
    u = uniform01_f32(seed, 3 * n).reshape(n, 3)
    which = (splitmix64(seed ^ 0xC1, n) % np.uint64(k)).astype(np.int64)
    face = (splitmix64(seed ^ 0xFA, n) % np.uint64(4)).astype(np.int64)
    w = np.float32(1.0)
    off = (u * np.float32(2.0) - np.float32(1.0)) * w
    p = ctr[which].copy()
    fx = face < 2
    p[:, 0] += np.where(fx, np.where(face == 0, -w, w), off[:, 0])
    p[:, 1] += np.where(fx, off[:, 1], np.where(face == 2, -w, w))
    p[:, 2] = np.float32(lo) + u[:, 2] * np.float32(min(8.0, hi - lo))
    return (np.round(p.astype(np.float64) / res) * res).astype(np.float32)
