#!/usr/bin/env python3
"""Where k_hploc_ext spends its time (measurement build: tools/build_variant.sh ext_t "-DABL_EXT_TIMING").
Usage (GPU box):  BVH_MI355X_LIB=build/variants/libbvh_ext_t.so python tools/ext_timing.py [N=10000000]"""
import ctypes as C, os, sys
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
names = ["tasks_tile", "passes", "passes_two_tasks", "tasks", "rounds(wave)", "cyc_rounds/64", "cyc_load/64", "cyc_handover/64", "wave_life/1024", "waves",
         "fast", "-", "slow_continue", "slow_stop"]
v = C.c_int64()
words = np.zeros((64, 32), np.int64)
for sub in range(64):
    for w in range(2, 32):
        assert pkg.lib().bvh_ctx_get_option(ctx.handle, 1000 + sub * 32 + w - 2, C.byref(v)) == 0
        words[sub, w] = v.value
tot = words.sum(0)
out = {nm: int(tot[2 + k]) for k, nm in enumerate(names)}
out["tasks_tile"] = int(words[0, 2])
print(n, out)
p = out["passes"]
print(f"passes {p}, two-task share {out['passes_two_tasks']/p:.2f}, tasks/pass {out['tasks']/p:.2f}, rounds/pass {out['rounds(wave)']/p:.2f}")
print(f"cycles per pass: load {out['cyc_load/64']*64/p:.0f}  rounds {out['cyc_rounds/64']*64/p:.0f} ({out['cyc_rounds/64']*64/max(1,out['rounds(wave)']):.0f} per round)  hand-over {out['cyc_handover/64']*64/p:.0f}")
print(f"waves {out['waves']}: mean life {out['wave_life/1024']*1024/max(1,out['waves']):.0f} cycles; sum of pass cycles / sum of wave life = {(out['cyc_load/64']+out['cyc_rounds/64']+out['cyc_handover/64'])*64/(out['wave_life/1024']*1024):.2f}")
print("wave-life histogram (bins of 32768 cycles):", tot[16:32].tolist())
