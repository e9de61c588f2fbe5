#!/usr/bin/env python3
"""Builds far beyond the benchmark size (the boundary accepts n < 2^30; one MI355X holds 288 GB): 50 M .. 400 M triangles generated ON the device
(no host copy of the mesh), every builder's two schedulers compared with each other (bvh_checksum: nodes + leaves + root), sorted keys checked on the
device, the root box compared with the scene extent, and — up to --validate-max leaves — the downloaded tree run through the oracle's validator.
What this is for: 32-bit offsets (n x 64 bytes of triangles passes 4 GB at 67 M), queue / status / arena sizing, and the level loops' counters at sizes
no test reaches.  Usage: python tools/large_n.py [--sizes 50000000 100000000 ...] [--validate-max 100000000] [--ploc-max 100000000]"""
import argparse, os, sys, time
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bvh_pkg
import oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", type=int, nargs="+", default=[50_000_000, 100_000_000, 200_000_000])
ap.add_argument("--validate-max", type=int, default=100_000_000)
ap.add_argument("--ploc-max", type=int, default=100_000_000)
ap.add_argument("--bits60-max", type=int, default=100_000_000)
a = ap.parse_args()
pkg = bvh_pkg.load(); ctx = pkg.Context(0)


class _Raw:                                   # a device address as a torch tensor (no copy)
    def __init__(self, ptr, n, typestr): self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_mesh(n, seed):
    """n small triangles with uniform random centres in the unit cube: 64-byte records {v0, v1, v2, pad}, written by torch on the device"""
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    t = torch.zeros((n, 16), device="cuda", dtype=torch.float32)
    step = 20_000_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        c = torch.rand((hi - lo, 3), device="cuda", generator=g)
        for v in range(3):
            t[lo:hi, 3 * v:3 * v + 3] = c + 0.001 * (torch.rand((hi - lo, 3), device="cuda", generator=g) - 0.5)
    return t


fails = 0
for n in a.sizes:
    t0 = time.time()
    d = device_mesh(n, 7); torch.cuda.synchronize()
    print(f"== n={n}: mesh on the device in {time.time() - t0:.1f} s ({n * 64 / 2**30:.1f} GiB)", flush=True)
    for algo, name, modes in ((3, "HPLOC", ("block", "async")), (1, "SinglePassLbvh", ("block", "single")), (0, "TwoPassLbvh", ("",)), (2, "PLOCNew", ("",))):
        if algo == 2 and n > a.ploc_max: continue
        sums = []
        for mode in modes:
            if algo == 3: ctx.set_option("hploc", mode)
            if algo == 1: ctx.set_option("lbvh", mode)
            b = pkg.BUILDERS[algo]()
            b.build(ctx, d, on_device=True, n=n)               # first build of this size: allocation
            ctx.set_profiling(1)                               # stage timers (the reference's TimerCodes)
            b.build(ctx, d, on_device=True, n=n)
            ctx.set_profiling(0)
            tm = dict(b.m_timer)
            chk = b.checksum(); sah = b.sah_cost()
            r = b.result
            keys = torch.as_tensor(_Raw(r.d_sorted_keys, n, "<u4"), device="cuda")
            # (uint32 has no comparison kernels in torch: compare as int64 in slices)
            srt = True
            for lo in range(0, n - 1, 50_000_000):
                hi = min(n - 1, lo + 50_000_000)
                k = keys[lo:hi + 1].to(torch.int64)
                srt = srt and bool((k[1:] >= k[:-1]).all())
            vals = torch.as_tensor(_Raw(r.d_sorted_vals, n, "<u4"), device="cuda")
            vsum = 0
            for lo in range(0, n, 50_000_000): vsum += int(vals[lo:lo + 50_000_000].to(torch.int64).sum())
            perm = vsum == n * (n - 1) // 2
            ok = srt and perm and np.isfinite(sah) and sah > 0 and r.n_leaves == n
            sums.append(chk)
            print(f"  {name:15s} {mode:6s} total {tm['TotalTime']:8.3f} ms  (E {tm['CalculateCentroidExtentsTime']:.3f} M {tm['CalculateMortonCodesTime']:.3f} S {tm['SortingTime']:.3f} B {tm['BvhBuildTime']:.3f})"
                  f"  {n / max(tm['TotalTime'], 1e-9) / 1e3:7.1f} Mtris/s  sah {sah:.4f}  root {r.root}  keys sorted {srt}  values a permutation (sum) {perm}  checksum {chk:016x}", flush=True)
            if not ok: fails += 1; print("  FAIL: property", flush=True)
            if n <= a.validate_max and mode == modes[0]:
                t1 = time.time()
                got = b.download()
                rc = orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"])
                sc = got["scene"][0]; root = got["nodes"][got["root"]]
                same_box = bool(np.array_equal(root["min"], sc["min"]) and np.array_equal(root["max"], sc["max"]))
                print(f"    oracle validator: {'ok' if rc == 0 else 'INVALID rc=%d' % rc}  root box == scene extent: {same_box}  ({time.time() - t1:.1f} s)", flush=True)
                if rc != 0 or same_box is False: fails += 1
                del got
            del b
        if len(sums) == 2 and sums[0] != sums[1]:
            fails += 1; print(f"  FAIL: {name} schedulers disagree", flush=True)
    if n <= a.bits60_max:
        ctx.set_option("hploc", "auto")
        b = pkg.HPLOC(); ctx.set_profiling(1); b.build_ex(ctx, n, tris=d, morton_bits=60); ctx.set_profiling(0)
        chk = b.checksum(); sah = b.sah_cost()
        print(f"  HPLOC 60-bit keys: total {b.m_timer['TotalTime']:.3f} ms  sah {sah:.4f}  checksum {chk:016x}", flush=True)
        del b
    ctx.set_option("hploc", "auto"); ctx.set_option("lbvh", "auto")
    del d; torch.cuda.empty_cache()
print(f"large_n: {fails} failures", flush=True)
sys.exit(1 if fails else 0)
