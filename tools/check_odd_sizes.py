import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, bvh_pkg, oracle as orc
pkg = bvh_pkg.load(); ctx = pkg.Context(0)
for n, kind in ((3_000_017, "uniform"), (1_234_567, "sponza"), (600_001, "bunny")):
    tris = pkg.meshgen.uniform(n, 5) if kind == "uniform" else pkg.meshgen.sponza_like(n, 3) if kind == "sponza" else pkg.meshgen.bunny_like(n, 2)
    n = len(tris)
    fe = orc.front_end(tris)
    for algo in (0, 1, 3, 2):
        b = pkg.BUILDERS[algo]().build(ctx, tris); got = b.download()
        ok = orc.validate_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"]) == 0 and np.array_equal(got["sorted_keys"], fe["skeys"]) and np.array_equal(got["sorted_vals"], fe["svals"])
        extra = ""
        if algo == 1:
            ref, root = orc.lbvh_single(tris, fe["skeys"], fe["svals"]); extra = f"bit-exact={got['nodes'].tobytes() == ref.tobytes() and root == got['root']}"
        elif algo == 0:
            ref, _ = orc.lbvh_two(tris, fe["skeys"], fe["svals"]); extra = f"bit-exact={got['nodes'].tobytes() == ref.tobytes()}"
        elif algo == 3 and n < 1_500_000:
            hn, hl, _ = orc.hploc(fe["boxes"], fe["skeys"], fe["svals"]); extra = f"topology={orc.topology_hash(got['nodes'], got['leaves'], 0, n, 1) == orc.topology_hash(hn, hl, 0, n, 1)}"
        print(n, kind, pkg.ALGO_NAMES[algo], "valid+sorted", ok, extra, flush=True)
