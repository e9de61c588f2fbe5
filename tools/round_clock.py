#!/usr/bin/env python3
"""How long one PLOC round takes a wave INSIDE k_hploc_block, and how much of a wave's level loop is rounds (measurement build: tools/build_variant.sh rc<k> "-DABL_ROUND_CLOCK
[-DABL_LDS_PAD=bytes]"; the pad limits the workgroups a CU holds).  The kernel's waves add {wave-rounds, shader-clock ticks inside rounds, active halves, ticks of the whole
level loop, waves} to words 20..24 of the sub-queues' padded heads.  python tools/round_clock.py [N]"""
import os, sys
import numpy as np, torch
torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
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
w = [sum(ctx.get_option(1000 + sq * 32 + 20 - 2 + k) for sq in range(64)) for k in range(5)]
rounds, ticks, halves, loop_ticks, waves = w
# u32 sums may wrap: ticks are summed per sub-queue in 32 bits (19 532 tiles x 4 waves x ~80 k ticks / 64 sub-queues ~ 1e8: fits)
print(f"lib={os.path.basename(os.environ.get('BVH_MI355X_LIB', 'main'))} n={n}: k_hploc_block {kms * 1e3:.1f} us; waves {waves}, wave-rounds {rounds} ({rounds / max(waves, 1):.2f} per wave), "
      f"active halves per wave-round {halves / max(rounds, 1):.3f}")
print(f"  ticks per wave-round {ticks / max(rounds, 1):.0f}; level loop per wave {loop_ticks / max(waves, 1):.0f} ticks, of which rounds {100.0 * ticks / max(loop_ticks, 1):.1f} %")
cyc = kms * 1e-3 * 2.4e9
print(f"  (at 2.4 GHz the launch is {cyc:.3e} cycles; waves inside a round per SIMD, launch average: {ticks / cyc / 1024:.2f}; waves inside the level loop per SIMD: {loop_ticks / cyc / 1024:.2f})", flush=True)
