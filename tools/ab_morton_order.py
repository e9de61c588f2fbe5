#!/usr/bin/env python3
"""k_extents / k_morton / k_hist per-kernel event times for the uniform mesh in random order and re-ordered by its own Morton order (a real mesh's usual state), plus the
stand-alone sort on sorted keys; the library comes from BVH_MI355X_LIB (tools/build_variant.sh).  python tools/ab_morton_order.py [N]"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0); L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tris = pkg.meshgen.uniform(n, 1)
got = pkg.BUILDERS[pkg.ALGO_SINGLEPASS]().build(ctx, tris).download()
order = got["sorted_vals"]; skeys = got["sorted_keys"].copy(); del got
for label, t in (("random", tris), ("morton", np.ascontiguousarray(tris[order]))):
    d = torch.from_numpy(t.view(np.uint8).reshape(-1)).cuda()
    for bits in (30, 60):
        b = pkg.HPLOC()
        for _ in range(3): b.build_ex(ctx, n, tris=d, morton_bits=bits)
        chk = b.checksum()
        ctx.set_profiling(2)
        for _ in range(20): b.build_ex(ctx, n, tris=d, morton_bits=bits)
        kt = ctx.kernel_times(); ctx.set_profiling(0)
        print(f"{label:6s} order, {bits}-bit keys: " + "  ".join(f"{k} {v[0] / 20:.4f}" for k, v in kt.items() if k in ("k_extents", "k_morton", "k_morton64", "k_onesweep")) + f"  checksum {chk:016x}", flush=True)
    del d
for label, k in (("random", np.random.default_rng(1).integers(0, 1 << 30, n, dtype=np.uint32)), ("sorted", skeys)):
    d_k = ctx.upload(k); d_sk = ctx.alloc(n * 4); d_sv = ctx.alloc(n * 4)
    for _ in range(3): assert L.bvh_sort_pairs(ctx.handle, d_k.ptr, None, n, d_sk.ptr, d_sv.ptr, 0, 30) == 0
    import time
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): assert L.bvh_sort_pairs(ctx.handle, d_k.ptr, None, n, d_sk.ptr, d_sv.ptr, 0, 30) == 0
    ctx.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
    ok = bool(np.array_equal(d_sk.download(np.uint32, n), np.sort(k, kind="stable")))
    print(f"stand-alone sort (k_hist + 4 passes), {label} keys: wall {wall:.4f} ms  sorted {ok}", flush=True)
    d_k.free(); d_sk.free(); d_sv.free()
