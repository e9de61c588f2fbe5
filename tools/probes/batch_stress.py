"""stress of the batched builder's multi-lane path (debugging aid): LD_PRELOAD=build/libsegv_bt.so python tools/probes/batch_stress.py [algos=0123] [sizes=sb] [keep=0|1|2] [iters]
algos: digits of bvh_algo; sizes: s = three ~5 k meshes, b = six 300 k meshes; keep: 0 never, 1 always, 2 alternate"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bvh_pkg
pkg = bvh_pkg.load()
algos = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "1302")]
sizes = sys.argv[2] if len(sys.argv) > 2 else "sb"
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 12
small = [pkg.meshgen.uniform(5000 + 11 * m, 7 + m) for m in range(3)]
big = [pkg.meshgen.uniform(300_000, 50 + m) for m in range(6)]
ref = {}
for it in range(iters):
    for algo in algos:
        b = pkg.Batch((0,))
        for tag, meshes in (("s", small), ("b", big)):
            if tag in sizes:
                r = b.build(meshes, algo, checksums=True, keep=(keep == 1 or (keep == 2 and it % 2 == 0)))
                key = (algo, tag)
                if key in ref:
                    assert np.array_equal(ref[key], r["checksums"]), (key, it)
                ref[key] = r["checksums"].copy()
        b.close()
    print("iteration", it, "ok", flush=True)
print("stress passed:", sys.argv[1:])
