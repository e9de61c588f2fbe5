#!/usr/bin/env python3
"""Timeline of the large tasks (> n / 4096 leaves) of k_hploc_ext (measurement build: tools/build_variant.sh tr0 "-DABL_EXT_TRACE").
Usage (GPU box):  BVH_MI355X_LIB=build/variants/libbvh_tr0.so python tools/ext_trace.py [N=10000000]"""
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
lib = pkg.lib()
lib.bvh_debug_read_queue.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
cap = C.c_uint64(); assert lib.bvh_debug_read_queue(ctx.handle, 2, 0, C.byref(cap), 1) == 0
cnt = np.zeros(1, np.uint32); assert lib.bvh_debug_read_queue(ctx.handle, 0, 34, cnt.ctypes.data, 1) == 0
q_cap = cap.value // 64
m = int(cnt[0])
m = min(m, q_cap // 8)
tr = np.zeros((m, 4), np.uint64); assert lib.bvh_debug_read_queue(ctx.handle, 1, q_cap * 63 + q_cap // 2, tr.ctypes.data, m * 4) == 0
t_start = int(tr[tr[:, 2] == 0][0, 0])
ev = tr[tr[:, 2] != 0]
t0 = (ev[:, 0].astype(np.int64) - t_start) / 100.0; t1 = (ev[:, 1].astype(np.int64) - t_start) / 100.0      # us (100 MHz clock)
L = (ev[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64); R = (ev[:, 2] >> np.uint64(32)).astype(np.int64)
kind = (ev[:, 3] & np.uint64(0xFF)).astype(int); rounds = ((ev[:, 3] >> np.uint64(8)) & np.uint64(0xFF)).astype(int)
d_load = ((ev[:, 3] >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64) / 100.0       # us from the pass's start until the work list is in registers
d_rnd = ((ev[:, 3] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64) / 100.0        # ... until the rounds are done
d_done = ((ev[:, 3] >> np.uint64(48)) & np.uint64(0xFFFF)).astype(np.int64) / 100.0       # ... until the hand-over is done (before the trace's own atomic)
size = R - L + 1
print(f"n={n}: {len(ev)} traced tasks; first start {t0.min():.1f} us, last end {t1.max():.1f} us")
names = {1: "fast(late)", 2: "slow-continue", 3: "stop", 4: "fast(early)"}
# the chain that ends at the root: follow it downwards (the child whose task ended last)
order = np.argsort(-size)
print("size-class summary (log2 size): count, mean duration us, mean rounds, first start, last end, kinds")
for lg in range(int(np.log2(n)) , int(np.log2(n / 4096)) - 1, -1):
    sel = (size > (1 << lg)) & (size <= (1 << (lg + 1)))
    if sel.any():
        ks = {names[k]: int((kind[sel] == k).sum()) for k in names if (kind[sel] == k).any()}
        print(f"  2^{lg}..: {sel.sum():5d}  dur {np.mean(t1[sel]-t0[sel]):5.2f}  rounds {rounds[sel].mean():.1f}  start {t0[sel].min():6.1f}  end {t1[sel].max():6.1f}  {ks}")
# critical chain: start from the root task, repeatedly pick the traced child range that ended last
root = np.argmax(size)
cur = root; chain = []
while True:
    chain.append(cur)
    inside = np.where((L >= L[cur]) & (R <= R[cur]) & (size < size[cur]))[0]
    if len(inside) == 0: break
    # direct children: maximal ranges inside
    mx = inside[np.argsort(-size[inside])][:2]
    kids = [k for k in mx if (L[k] == L[cur] or R[k] == R[cur])]
    if not kids: break
    cur = max(kids, key=lambda k: t1[k])
print("critical chain, bottom-up: size, start, end, rounds, hand-over kind; then the level's phases in us")
prev_end = None
for c in reversed(chain):
    gap = (t0[c] - prev_end) if prev_end is not None else 0.0
    print(f"  {size[c]:9d}  {t0[c]:7.1f} -> {t1[c]:7.1f}  ({t1[c]-t0[c]:5.2f} us, {rounds[c]} rounds)  {names.get(kind[c], kind[c])}   gap before {gap:5.2f}  list load {d_load[c]:5.2f}  rounds {d_rnd[c]-d_load[c]:5.2f}  hand-over {d_done[c]-d_rnd[c]:5.2f}  (the trace's own atomic {t1[c]-t0[c]-d_done[c]:5.2f})")
    prev_end = t1[c]
