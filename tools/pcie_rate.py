#!/usr/bin/env python3
"""PCIe-inclusive build rate (host Triangle[] in, as the reference's X::build(Context&, std::vector<Triangle>&) is called):
H2D copy of the 64-byte records + build.  Reported in DESIGN.md §6; never the bench `value`."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bvh_pkg
pkg = bvh_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tris = pkg.meshgen.uniform(n, 1)
ctx = pkg.Context(0); ctx.reserve(n)
b = pkg.HPLOC()
for _ in range(2):
    b.build(ctx, tris); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    b.build(ctx, tris); ctx.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"host-input HPLOC build of {n} triangles (pageable numpy memory): {dt*1e3:.2f} ms  -> {n/dt/1e6:.1f} Mtris/s  ({n*64/dt/1e9:.1f} GB/s incl. build)")
