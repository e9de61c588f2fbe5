#!/usr/bin/env python3
"""stage S alone (bvh_sort_pairs on random 30-bit keys, 10 M) for kernel traces of the BVH_SORT_DEBUG ablations"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0); L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
end_bit = int(sys.argv[2]) if len(sys.argv) > 2 else 30          # 30: the build's 8/8/8/6 passes; 32: four 8-bit passes
keys = np.random.default_rng(1).integers(0, 1 << end_bit, n, dtype=np.uint32)
d_k = ctx.upload(keys); d_sk = ctx.alloc(n * 4); d_sv = ctx.alloc(n * 4)
for _ in range(10): assert L.bvh_sort_pairs(ctx.handle, d_k.ptr, None, n, d_sk.ptr, d_sv.ptr, 0, end_bit) == 0
ctx.synchronize()
if not (int(os.environ.get("BVH_SORT_DEBUG", "0")) & 131):        # (bits 1 / 2 / 128 are timing ablations with wrong results)
    order = np.argsort(keys, kind="stable").astype(np.uint32)
    sk = d_sk.download(np.uint32, n); sv = d_sv.download(np.uint32, n)
    print("sorted == stable argsort:", bool(np.array_equal(sv, order) and np.array_equal(sk, keys[order])), flush=True)
reps = 30
ctx.set_profiling(2)
import time
t0 = time.perf_counter()
for _ in range(reps): assert L.bvh_sort_pairs(ctx.handle, d_k.ptr, None, n, d_sk.ptr, d_sv.ptr, 0, end_bit) == 0
ctx.synchronize(); dt = time.perf_counter() - t0
kt = ctx.kernel_times()
print(f"n={n} end_bit={end_bit} dbg={os.environ.get('BVH_SORT_DEBUG', '0')}: " + "  ".join(f"{k} {v[0] / reps:.4f} ms ({v[1] // reps} launches)" for k, v in kt.items()) + f"  wall {dt / reps * 1e3:.4f} ms", flush=True)
