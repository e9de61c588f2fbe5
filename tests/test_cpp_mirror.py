"""The C++ host mirror (include/bvh/builders.hpp) RUN on the GPU: examples/reference_driver.cpp is the body of the reference's driver
(src/main.cpp:52-77 — `X bvh; bvh.build(context, triangles); bvh.traverseBvh(context);`) for all four builders; its trees, public members,
BVH4 cost and image are diffed with the ctypes path and the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

ALGO = {"two": 0, "single": 1, "ploc": 2, "hploc": 3}
# what each reference traverseBvh() renders with (src/TwoPassLbvh.cpp:199-311: speculative while-while, t=(0,0,-5) s=1;
# src/SinglePassLbvh.cpp:190-311: if-if, t=(0,0,-3) s=3; PLOC++ / HPLOC print only)
FLAVOUR = {"two": (3, (0.0, 0.0, -5.0), 1.0), "single": (2, (0.0, 0.0, -3.0), 3.0)}


@pytest.fixture(scope="module")
def driver(tmp_path_factory, pkg):
    out = str(tmp_path_factory.mktemp("drv") / "reference_driver")
    libdir = os.path.dirname(pkg.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "reference_driver.cpp"), "-o", out,
           "-L", libdir, "-lbvh_mi355x", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def read_dump(path):
    raw = open(path, "rb").read(); off = 0; secs = []
    while off < len(raw):
        n, sz = struct.unpack_from("<QQ", raw, off); off += 16
        secs.append((n, sz, raw[off: off + n * sz])); off += n * sz
    return secs


@pytest.fixture(scope="module")
def mesh_files(tmp_path_factory, pkg):
    d = tmp_path_factory.mktemp("meshes")
    out = {"cornell382": os.path.join(GOLDEN, "cornell382.tri")}
    t = pkg.meshgen.uniform(30_000, 21)
    p = str(d / "uniform30k.tri")
    np.stack([t["v1"], t["v2"], t["v3"]], axis=1).astype("<f4").tofile(p)
    out["uniform30k"] = p
    return out


@pytest.mark.parametrize("mesh", ["cornell382", "uniform30k"])
@pytest.mark.parametrize("which", ["two", "single", "ploc", "hploc"])
def test_reference_driver_runs_and_matches_ctypes_path(pkg, orc, ctx, driver, mesh_files, tmp_path, which, mesh):
    dump = str(tmp_path / "dump.bin")
    r = subprocess.run([driver, which, mesh_files[mesh], dump], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # the reference's perf block (src/TwoPassLbvh.cpp:300-310)
    assert "kept sort stage: ok" in r.stdout      # d_mortonCodeValues.ptr() handed to a kept Oro::RadixSort::sort call site (src/Hploc.cpp:63-81) is a real device array
    for token in ("Perf Times", "CalculateCentroidExtentsTime :", "CalculateMortonCodesTime :", "SortingTime : ", "BvhBuildTime : ", "CollapseTime : ", "Bvh Cost : ", "Total Time : "):
        assert token in r.stdout
    tris = pkg.meshgen.load_tri(mesh_files[mesh]); n = len(tris)
    algo = ALGO[which]
    head, nodes, leaves, skeys, svals, keys, vals, boxes, scene, tbuf, wide, wleaves, image = read_dump(dump)
    head = np.frombuffer(head[2], dtype=np.float32)
    b = pkg.BUILDERS[algo]().build(ctx, tris); got = b.download()
    # same trees as the ctypes path, byte for byte (every builder's numbering is schedule independent here)
    assert nodes[2] == got["nodes"].tobytes() and int(head[0]) == got["root"] and int(head[1]) == n - 1
    assert leaves[2] == (got["leaves"].tobytes() if got["leaves"] is not None else b"")
    assert skeys[2] == got["sorted_keys"].tobytes() and svals[2] == got["sorted_vals"].tobytes()
    # the reference's remaining public members: d_mortonCodeKeys / Values, d_triangleAabb, d_sceneExtents, d_triangleBuff
    ob, oscene = orc.prim_bounds(tris); okeys, ovals = orc.morton_codes(ob, oscene)
    assert keys[2] == okeys.tobytes() and vals[2] == ovals.tobytes() and boxes[2] == ob.tobytes() and scene[2] == oscene.tobytes()
    assert tbuf[2] == tris.tobytes()
    # m_cost = Utility::calculatebvh4Cost of the collapsed tree (src/TwoPassLbvh.cpp:196)
    w = np.frombuffer(wide[2], dtype=pkg.BVH4_NODE); p = np.frombuffer(wleaves[2], dtype=pkg.PRIM_NODE)
    assert len(w) == int(head[4]) and len(p) == n
    ow, opn, ototal = orc.collapse4(got["nodes"], got["leaves"], got["root"], n, got["layout"])
    assert len(w) == ototal and orc.topology_hash4(w, p, len(w), n) == orc.topology_hash4(ow, opn, ototal, n)
    c64, c32 = orc.sah_bvh4(ow, opn, ob, ototal, n)
    assert head[2] == pytest.approx(c64, rel=2e-6) and head[3] == pytest.approx(orc.sah_bvh2(got["nodes"], got["leaves"], got["root"], n, got["layout"])[0], rel=2e-6)
    R = orc.ref_utility()
    if R is not None:
        wc = np.ascontiguousarray(w); pc = np.ascontiguousarray(p)
        c_ref = R.ref_calculatebvh4Cost(wc.ctypes.data, pc.ctypes.data, ob.ctypes.data, 0, len(w), n - 1)       # f32 accumulation in node order
        assert c_ref == pytest.approx(orc.sah_bvh4(w, p, ob, len(w), n)[1], rel=1e-6) and head[2] == pytest.approx(c_ref, rel=1e-3)
    assert head[5] > 0 and head[6] > 0 and head[7] > 0 and head[8] > 0 and head[9] > 0       # Timer tokens incl. CollapseBvhTime
    # traverseBvh's image
    if which in FLAVOUR:
        kind, tr, sc = FLAVOUR[which]
        cam, xf = pkg.cornell_view(); xf["translation"][0] = tr; xf["scale"][0] = (sc, sc, sc)
        img, _ = b.render(tris, cam, xf, 512, kind=kind)
        assert int(head[10]) == 512 and int(head[11]) == 512 and image[2] == img.tobytes()
        if mesh == "cornell382":
            assert img[3::4].sum() > 255 * 1000, "the reference's view of the Cornell box must see geometry"
    else:
        assert image[0] == 0


def test_batched_builder_mirror(driver, mesh_files):
    r = subprocess.run([driver, "batched", mesh_files["uniform30k"]], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all-gather" in r.stdout and r.stdout.count("mesh ") == 4
    # the reference's public members (src/BatchedBuilder.h:24-30: d_bvhNodes / d_primRefs / d_rootNodes, m_rootNodeIdx, m_timer, m_nInternalNodes, m_cost) are read by
    # the driver itself (it exits non-zero if a root node is not where d_rootNodes says); four 30 000-triangle HPLOC meshes on one device: 4 x 29 999 nodes, 3 lanes
    assert "batched: 119996 nodes, 120000 leaves" in r.stdout and "lanes per device" in r.stdout
