cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
echo "== $v"
BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so timeout 120 python - <<'PY'
import os, sys
import numpy as np, torch
torch.cuda.init()
sys.path.insert(0, "/root/repo")
import bvh_pkg
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for n in (10_000_000, 2_000_000):
    tris = pkg.meshgen.uniform(n, 1)
    d = torch.from_numpy(tris.view(np.uint8).reshape(-1)).cuda()
    ctx.set_option("hploc", "block")
    b = pkg.HPLOC()
    for _ in range(3): b.build(ctx, d, on_device=True, n=n)
    ctx.set_profiling(2)
    for _ in range(20): b.build(ctx, d, on_device=True, n=n)
    kt = ctx.kernel_times(); ctx.set_profiling(0)
    print(n, "  ".join(f"{k} {v[0]/20:.4f}" for k, v in kt.items() if "hploc" in k), flush=True)
PY
done
