"""Synthetic triangle meshes for the BVH build benchmarks and parity tests.

The reference's bunny / sponza OBJ files are not in its tree (``.MISSING_LARGE_BLOBS``), so the big configurations run
on generated stand-ins.  All generators are counter based (SplitMix64 of ``seed ^ counter``): any slice of a mesh can be
regenerated independently, and the same code produces the same bytes on every machine.

Every generator returns a C-contiguous numpy array of dtype ``TRIANGLE`` (64-byte records, reference
``src/Common.h:429-434``).
"""
from __future__ import annotations

import numpy as np

TRIANGLE = np.dtype([("v1", "<f4", 3), ("v2", "<f4", 3), ("v3", "<f4", 3), ("pad", "<f4", 7)])
assert TRIANGLE.itemsize == 64

_U64 = np.uint64


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """One SplitMix64 output for every 64-bit state in ``x`` (vectorised, wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _unit(seed: int, counter: np.ndarray, lane: int) -> np.ndarray:
    """U[0,1) float32 with 24 random bits: top 24 bits of SplitMix64(seed ^ (counter*4 + lane))."""
    with np.errstate(over="ignore"):
        state = _U64(seed) ^ (counter.astype(np.uint64) * _U64(4) + _U64(lane))
    bits = _splitmix64(state) >> _U64(40)
    return (bits.astype(np.float32) / np.float32(16777216.0)).astype(np.float32)


def _pack(v1: np.ndarray, v2: np.ndarray, v3: np.ndarray) -> np.ndarray:
    out = np.zeros(v1.shape[0], dtype=TRIANGLE)
    out["v1"], out["v2"], out["v3"] = v1, v2, v3
    return out


def uniform(n: int, seed: int = 1, offset=(0.0, 0.0, 0.0), start: int = 0) -> np.ndarray:
    """``n`` small triangles with centres ~ U[0,1)^3, edge scale s = 2 * n_total^(-1/3) (config 3 / 5 workload).

    ``start`` regenerates the slice [start, start+n) of a larger mesh (the scale then still uses ``n`` — pass the same
    ``n`` for all slices only through :func:`uniform_slice`).
    """
    return uniform_slice(n, n, seed, offset, start)


def uniform_slice(n_total: int, count: int, seed: int, offset=(0.0, 0.0, 0.0), start: int = 0) -> np.ndarray:
    i = np.arange(start, start + count, dtype=np.uint64)
    s = np.float32(2.0 * float(n_total) ** (-1.0 / 3.0))
    off = np.asarray(offset, dtype=np.float32)
    # counters: point p of triangle i uses counter i*16 + p*3 + axis  (p = 0 centre, 1..3 vertices)
    def vec(p: int) -> np.ndarray:
        return np.stack([_unit(seed, i * _U64(16) + _U64(p * 3 + a), 0) for a in range(3)], axis=1)
    c = vec(0) + off
    half = np.float32(0.5)
    v = [(c + s * (vec(p) - half)).astype(np.float32) for p in (1, 2, 3)]
    return _pack(*v)


def probe_mesh(n: int) -> np.ndarray:
    """The survey's probe generator, bit for bit: std::mt19937 rng(1234) and
    std::uniform_real_distribution<float>(0, 1); per triangle c = (U,U,U), then three vertices c + (U-.5)*.01 per component (draw order
    c.x c.y c.z, then x y z of each vertex).  numpy's RandomState(1234) seeds MT19937 with init_genrand like std::mt19937 does; libstdc++'s
    uniform_real_distribution<float> draws ONE 32-bit word per value: float(word) / 2^32 (round to nearest), and 1.0 becomes nextafter(1, 0)
    (std::generate_canonical<float, 24>).  Pins SURVEY.md §8(c)'s known answers (tests/test_oracle_golden.py)."""
    raw = np.random.RandomState(1234)._bit_generator.random_raw(12 * n).astype(np.uint32)
    u = raw.astype(np.float32) / np.float32(4294967296.0)
    u = np.where(u >= np.float32(1.0), np.nextafter(np.float32(1.0), np.float32(0.0)), u).astype(np.float32).reshape(n, 12)
    c = u[:, 0:3]
    v = [(c + (u[:, 3 + 3 * k: 6 + 3 * k] - np.float32(0.5)) * np.float32(0.01)).astype(np.float32) for k in range(3)]
    return _pack(*v)


def np_mt_mesh(n: int) -> np.ndarray:
    """Same shape of mesh from numpy's Generator(MT19937(1234)) doubles (round 1's stand-in for the probe generator; kept because the
    committed reference-kernel outputs of the golden mesh "probe5000" were produced on it)."""
    rng = np.random.Generator(np.random.MT19937(1234))
    u = rng.random((n, 12)).astype(np.float32)
    c = u[:, 0:3]
    v = [(c + (u[:, 3 + 3 * k: 6 + 3 * k] - np.float32(0.5)) * np.float32(0.01)).astype(np.float32) for k in range(3)]
    return _pack(*v)


def _icosphere_faces(min_faces: int):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                      [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    verts /= np.linalg.norm(verts, axis=1, keepdims=True)
    faces = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                      [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                      [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    tri = verts[faces]                      # (F,3,3)
    while tri.shape[0] < min_faces:
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = (a + b) / 2, (b + c) / 2, (c + a) / 2
        for m in (ab, bc, ca):
            m /= np.linalg.norm(m, axis=1, keepdims=True)
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)], 0)
    return tri


def bunny_like(n: int = 150_000, seed: int = 2) -> np.ndarray:
    """Closed, roughly uniformly tessellated blob: icosphere with 3 octaves of hash noise on the radius (Bunny stand-in)."""
    tri = _icosphere_faces(n)[:n]           # (n,3,3) unit directions
    d = tri.reshape(-1, 3)
    r = np.ones(d.shape[0], dtype=np.float64)
    for octave, (freq, amp) in enumerate(((3.0, 0.18), (7.0, 0.08), (17.0, 0.03))):
        cell = np.floor((d + 2.0) * freq).astype(np.int64)
        h = (cell[:, 0] * 73856093) ^ (cell[:, 1] * 19349663) ^ (cell[:, 2] * 83492791) ^ (seed * 1000003 + octave)
        u = (_splitmix64(h.astype(np.uint64)) >> _U64(40)).astype(np.float64) / 16777216.0
        r += amp * (u - 0.5)
    p = (d * r[:, None] * 0.8 + np.array([0.0, 0.9, 0.0])).astype(np.float32).reshape(-1, 3, 3)
    return _pack(p[:, 0], p[:, 1], p[:, 2])


def sponza_like(n: int = 262_144, seed: int = 3) -> np.ndarray:
    """Axis-aligned room 30x12x18: 20 % large wall/floor triangles (exactly axis aligned -> zero-extent AABBs), 80 % small
    triangles clustered around columns/drapes — Sponza's size variance and axis alignment (stresses LBVH vs PLOC SAH)."""
    n_big = n // 5
    n_small = n - n_big
    i = np.arange(n_big, dtype=np.uint64)
    room = np.array([30.0, 12.0, 18.0], dtype=np.float32)
    # big: quads on the 6 walls split in two triangles; wall = i % 6, cell grid on the wall
    wall = (i % _U64(6)).astype(np.int64)
    axis = wall // 2
    side = (wall % 2).astype(np.float32)
    u0 = _unit(seed, i, 0); v0 = _unit(seed, i, 1); su = _unit(seed, i, 2) * np.float32(0.12) + np.float32(0.02); sv = _unit(seed, i, 3) * np.float32(0.12) + np.float32(0.02)
    flip = ((i >> _U64(3)) & _U64(1)).astype(bool)
    def on_wall(uu, vv):
        p = np.zeros((n_big, 3), dtype=np.float32)
        a1 = (axis + 1) % 3; a2 = (axis + 2) % 3
        rows = np.arange(n_big)
        p[rows, axis] = side * room[axis]
        p[rows, a1] = np.clip(uu, 0, 1) * room[a1]
        p[rows, a2] = np.clip(vv, 0, 1) * room[a2]
        return p
    A = on_wall(u0, v0); B = on_wall(u0 + su, v0); C = on_wall(u0 + su, v0 + sv); D = on_wall(u0, v0 + sv)
    big = _pack(A, np.where(flip[:, None], C, B), np.where(flip[:, None], D, C))
    # small: 64 columns; triangles jittered around a column axis
    j = np.arange(n_small, dtype=np.uint64) + _U64(1 << 40)
    col = (_splitmix64(j ^ _U64(seed * 7919)) % _U64(64)).astype(np.int64)
    cx = (col % 8).astype(np.float32) * np.float32(30.0 / 8) + np.float32(1.8)
    cz = (col // 8).astype(np.float32) * np.float32(18.0 / 8) + np.float32(1.1)
    ang = _unit(seed, j, 0) * np.float32(2 * np.pi)
    rad = np.float32(0.35) + _unit(seed, j, 1) * np.float32(0.1)
    h = _unit(seed, j, 2) * room[1]
    base = np.stack([cx + rad * np.cos(ang), h, cz + rad * np.sin(ang)], axis=1).astype(np.float32)
    e = np.float32(0.06)
    def jit(l0):
        return np.stack([(_unit(seed, j, 3) if l0 == 0 else _unit(seed ^ (0x1111 * l0), j, 3)) - np.float32(0.5),
                         _unit(seed ^ (0x2222 * (l0 + 1)), j, 1) - np.float32(0.5),
                         _unit(seed ^ (0x3333 * (l0 + 1)), j, 2) - np.float32(0.5)], axis=1).astype(np.float32) * e
    small = _pack(base + jit(0), base + jit(1), base + jit(2))
    out = np.concatenate([big, small])
    # interleave deterministically so that primitive order is not sorted by size
    perm = np.argsort(_splitmix64(np.arange(n, dtype=np.uint64) ^ _U64(seed)), kind="stable")
    return np.ascontiguousarray(out[perm])


def load_tri(path: str) -> np.ndarray:
    """Raw little-endian float32 triangle list: n x 9 floats (v1, v2, v3) as written by tools/make_golden.py."""
    raw = np.fromfile(path, dtype="<f4").reshape(-1, 3, 3)
    return _pack(raw[:, 0], raw[:, 1], raw[:, 2])
