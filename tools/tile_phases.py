#!/usr/bin/env python3
"""Where a tile of k_hploc_block spends its life (thread 0's view, the CU's shader clock: ticks are converted with the kernel's own event time and the in-flight count below is what calibrates it), on a build with -DABL_TILE_PHASES (tools/build_variant.sh phases "-DABL_TILE_PHASES").
python tools/tile_phases.py [N]"""
import os, sys
import numpy as np, torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
TICKS_PER_US = float(os.environ.get('TICKS_PER_US', '100'))     # s_memtime: 100 MHz on gfx950 (constant), not the shader clock
for n in [int(x) for x in sys.argv[1:]] or [10_000_000, 2_000_000]:
    tris = pkg.meshgen.uniform(n, 1)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    ctx.set_option("hploc", "block")
    b = pkg.HPLOC()
    for _ in range(3): b.build(ctx, d, on_device=True, n=n)
    ctx.set_profiling(2)
    for _ in range(5): b.build(ctx, d, on_device=True, n=n)
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    kms = kt["k_hploc_block"][0] / 5
    b.build(ctx, d, on_device=True, n=n)
    w = [sum(ctx.get_option(1000 + sq * 32 + 4 - 2 + k) for sq in range(64)) for k in range(11)]
    tiles = max(w[10], 1)
    names = ["staging (loads -> LDS, barrier)", "ranges + level sort", "level loop", "hand-over steps 1-2", "hand-over step 3 (queue)"]
    print(f"n={n}: {tiles} tiles; per tile, us:")
    tot = 0.0
    for k, nm in enumerate(names):
        us = w[k] / tiles / TICKS_PER_US; tot += us
        print(f"  {nm:34s} {us:7.2f}")
    print(f"  {'tile life':34s} {tot:7.2f}   (level loop: {w[9] / tiles:.2f} non-empty levels, {w[8] / tiles / TICKS_PER_US:.2f} us of it waiting at the levels' barriers)")
    print(f"  kernel {kms * 1e3:.1f} us (events, this build); sum of tile lives / kernel time = {tiles * tot / (kms * 1e3):.0f} tiles in flight on average ({tiles * tot / (kms * 1e3) / 256:.1f} per CU)", flush=True)
