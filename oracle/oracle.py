"""ctypes binding of the CPU oracle (oracle/libbvh_oracle.so) and of the reference builds under oracle/_ref/.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package.  See oracle/bvh_oracle.cpp for what each function restates (reference file:line).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libbvh_oracle.so")
REF_UTILITY = os.path.join(_HERE, "_ref", "libref_utility.so")
REF_DRIVER = os.path.join(_HERE, "_ref", "libref_driver.so")

AABB = np.dtype([("min", "<f4", 3), ("max", "<f4", 3)])
TRIANGLE = np.dtype([("v1", "<f4", 3), ("v2", "<f4", 3), ("v3", "<f4", 3), ("pad", "<f4", 7)])
BVH2_NODE = np.dtype([("left", "<u4"), ("right", "<u4"), ("min", "<f4", 3), ("max", "<f4", 3)])
PRIMREF = np.dtype([("prim", "<u4"), ("min", "<f4", 3), ("max", "<f4", 3)])
SAH_NODE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("first", "<u4"), ("count", "<u4")])
BVH4_NODE = np.dtype([("aabb", AABB, 4), ("child", "<u4", 4), ("parent", "<u4"), ("count", "<u4"), ("pad", "<u4", 2)])
PRIM_NODE = np.dtype([("prim", "<u4"), ("parent", "<u4")])
assert BVH4_NODE.itemsize == 128 and SAH_NODE.itemsize == 32
STATS = np.dtype([("iterations", "<u8"), ("cluster_loads", "<u8"), ("cluster_stores", "<u8"), ("merge_calls", "<u8"), ("nn_rounds", "<u8")])


def build(ref: bool = True) -> None:
    """make -C oracle (own restatement always; reference builds only where /root/reference exists)."""
    targets = ["libbvh_oracle.so"] + (["ref"] if ref else [])
    r = subprocess.run(["make", "-C", _HERE] + targets, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build(ref=False)
        L = C.CDLL(LIB)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.orc_prim_bounds.argtypes = [vp, u32, vp, vp]
        L.orc_primrefs.argtypes = [vp, u32, vp]
        L.orc_morton_plan.argtypes = [vp, vp]
        L.orc_morton_codes.argtypes = [vp, u32, u32, u32, vp, vp, vp]
        L.orc_sort_pairs.argtypes = [vp, vp, u32, vp, vp]
        L.orc_morton_codes64.argtypes = [vp, u32, u32, u32, vp, C.c_int, vp]
        L.orc_lbvh_single64.argtypes = [vp, u32, vp, vp, vp]; L.orc_lbvh_single64.restype = u32
        L.orc_lbvh_two64.argtypes = [vp, u32, vp, vp, vp, vp]
        L.orc_hploc64.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        L.orc_lbvh_single.argtypes = [vp, u32, vp, vp, vp]; L.orc_lbvh_single.restype = u32
        L.orc_lbvh_two.argtypes = [vp, u32, vp, vp, vp, vp]
        L.orc_ploc.argtypes = [vp, u32, vp, vp, vp, vp]
        L.orc_hploc.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        L.orc_sah_bvh2.argtypes = [vp, vp, u32, u32, C.c_int, C.POINTER(C.c_float)]; L.orc_sah_bvh2.restype = C.c_double
        L.orc_validate_bvh2.argtypes = [vp, vp, u32, u32, C.c_int]; L.orc_validate_bvh2.restype = C.c_int
        L.orc_topology_hash.argtypes = [vp, vp, u32, u32, C.c_int]; L.orc_topology_hash.restype = u64
        L.orc_fnv1a.argtypes = [vp, u64]; L.orc_fnv1a.restype = u64
        L.orc_ploc_to_lbvh_layout.argtypes = [vp, vp, u32, vp]
        L.orc_binned_sah_build.argtypes = [vp, u32, vp]; L.orc_binned_sah_build.restype = u32
        L.orc_sah_binned.argtypes = [vp, u32, u32, C.POINTER(C.c_float)]; L.orc_sah_binned.restype = C.c_double
        L.orc_collapse4.argtypes = [vp, vp, u32, u32, C.c_int, vp, vp]; L.orc_collapse4.restype = u32
        L.orc_topology_hash4.argtypes = [vp, vp, u32, u32]; L.orc_topology_hash4.restype = u64
        L.orc_generate_rays.argtypes = [vp, vp, u32, u32]
        L.orc_trace_while.argtypes = [vp, vp, vp, vp, vp, u32, u32, u32, u32, C.POINTER(u32)]
        L.orc_sah_bvh4.argtypes = [vp, vp, vp, u32, u32, C.POINTER(C.c_float)]; L.orc_sah_bvh4.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return None if a is None else np.ascontiguousarray(a).ctypes.data


# ---- stages -------------------------------------------------------------------------------------------------------
def prim_bounds(tris: np.ndarray):
    n = tris.shape[0]
    boxes = np.empty(n, dtype=AABB); scene = np.empty(1, dtype=AABB)
    lib().orc_prim_bounds(tris.ctypes.data, n, boxes.ctypes.data, scene.ctypes.data)
    return boxes, scene


def primrefs(tris: np.ndarray) -> np.ndarray:
    refs = np.empty(tris.shape[0], dtype=PRIMREF)
    lib().orc_primrefs(tris.ctypes.data, tris.shape[0], refs.ctypes.data)
    return refs


def morton_plan(scene: np.ndarray) -> dict:
    p = np.zeros(10, dtype=np.int32)
    lib().orc_morton_plan(scene.ctypes.data, p.ctypes.data)
    return {"axis": p[0:3].tolist(), "bits": p[3:6].tolist(), "pre": p[6:8].tolist(), "pre_sum": int(p[8]), "swap": int(p[9])}


def morton_codes(boxes: np.ndarray, scene: np.ndarray):
    n = boxes.shape[0]
    keys = np.empty(n, dtype=np.uint32); vals = np.empty(n, dtype=np.uint32)
    lib().orc_morton_codes(boxes.ctypes.data, boxes.dtype.itemsize, 0, n, scene.ctypes.data, keys.ctypes.data, vals.ctypes.data)
    return keys, vals


def morton_codes_with_plan(boxes: np.ndarray, scene: np.ndarray, plan) -> np.ndarray:
    """orc_morton_codes_plan: the oracle's encoder driven by a GIVEN per-scene plan (list of 10 ints: axis[3], bits[3], pre[2], pre_sum, swap)"""
    n = boxes.shape[0]
    keys = np.empty(n, dtype=np.uint32); pl = np.asarray(plan, dtype=np.int32)
    L = lib(); L.orc_morton_codes_plan.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_morton_codes_plan(boxes.ctypes.data, boxes.dtype.itemsize, 0, n, scene.ctypes.data, pl.ctypes.data, keys.ctypes.data)
    return keys


def morton_codes64(boxes: np.ndarray, scene: np.ndarray, total_bits: int = 60) -> np.ndarray:
    """u64 extended Morton codes with a total_bits budget (30 reproduces morton_codes)"""
    n = boxes.shape[0]
    keys = np.empty(n, dtype=np.uint64)
    lib().orc_morton_codes64(boxes.ctypes.data, boxes.dtype.itemsize, 0, n, scene.ctypes.data, total_bits, keys.ctypes.data)
    return keys


def sort_pairs(keys: np.ndarray, vals: np.ndarray | None = None):
    if keys.dtype == np.uint64:      # the adopted contract, on u64 keys: stable ascending
        order = np.argsort(keys, kind="stable").astype(np.uint32)
        return keys[order], (order if vals is None else vals[order])
    n = keys.shape[0]
    sk = np.empty(n, dtype=np.uint32); sv = np.empty(n, dtype=np.uint32)
    lib().orc_sort_pairs(keys.ctypes.data, _p(vals), n, sk.ctypes.data, sv.ctypes.data)
    return sk, sv


def front_end(tris: np.ndarray, morton_bits: int = 30):
    """E + M + S -> dict(boxes, scene, keys, skeys, svals); morton_bits 60: u64 keys"""
    boxes, scene = prim_bounds(tris)
    if morton_bits == 60:
        keys = morton_codes64(boxes, scene, 60); sk, sv = sort_pairs(keys)
    else:
        keys, vals = morton_codes(boxes, scene); sk, sv = sort_pairs(keys, vals)
    return {"boxes": boxes, "scene": scene, "keys": keys, "skeys": sk, "svals": sv}


# ---- emitters -----------------------------------------------------------------------------------------------------
def lbvh_single(tris, skeys, svals):
    n = tris.shape[0]
    nodes = np.zeros(2 * n - 1, dtype=BVH2_NODE)
    fn = lib().orc_lbvh_single64 if skeys.dtype == np.uint64 else lib().orc_lbvh_single
    root = fn(tris.ctypes.data, n, skeys.ctypes.data, svals.ctypes.data, nodes.ctypes.data)
    return nodes, int(root)


def lbvh_two(tris, skeys, svals):
    n = tris.shape[0]
    refs = primrefs(tris)
    nodes = np.zeros(2 * n - 1, dtype=BVH2_NODE)
    parents = np.empty(2 * n - 1, dtype=np.uint32)
    fn = lib().orc_lbvh_two64 if skeys.dtype == np.uint64 else lib().orc_lbvh_two
    fn(refs.ctypes.data, n, skeys.ctypes.data, svals.ctypes.data, nodes.ctypes.data, parents.ctypes.data)
    return nodes, parents


def ploc(boxes, svals):
    n = boxes.shape[0]
    nodes = np.zeros(n - 1, dtype=BVH2_NODE); leaves = np.zeros(n, dtype=PRIMREF); st = np.zeros(1, dtype=STATS)
    lib().orc_ploc(boxes.ctypes.data, n, svals.ctypes.data, nodes.ctypes.data, leaves.ctypes.data, st.ctypes.data)
    return nodes, leaves, {k: int(st[k][0]) for k in STATS.names}


def hploc(boxes, skeys, svals):
    n = boxes.shape[0]
    nodes = np.zeros(n - 1, dtype=BVH2_NODE); leaves = np.zeros(n, dtype=PRIMREF); st = np.zeros(1, dtype=STATS)
    fn = lib().orc_hploc64 if skeys.dtype == np.uint64 else lib().orc_hploc
    fn(boxes.ctypes.data, n, skeys.ctypes.data, svals.ctypes.data, nodes.ctypes.data, leaves.ctypes.data, st.ctypes.data)
    return nodes, leaves, {k: int(st[k][0]) for k in STATS.names}


def hploc_tasks(boxes, skeys, svals) -> np.ndarray:
    """the merge tasks of the HPLOC build of these sorted leaves: (T, 5) uint32 {L, R, split, rounds, clusters loaded} in the oracle's execution order"""
    n = boxes.shape[0]
    L = lib(); L.orc_hploc_tasks.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]; L.orc_hploc_tasks.restype = C.c_uint32
    cap = n // 8 + 16
    out = np.zeros((cap, 5), dtype=np.uint32)
    cnt = L.orc_hploc_tasks(boxes.ctypes.data, n, skeys.ctypes.data, svals.ctypes.data, out.ctypes.data, cap)
    assert cnt <= cap
    return out[:cnt]


def build_tree(algo: int, tris: np.ndarray, morton_bits: int = 30) -> dict:
    """Whole pipeline on the CPU.  algo: 0 two-pass, 1 single-pass, 2 PLOC++, 3 HPLOC."""
    fe = front_end(tris, morton_bits)
    out = dict(fe)
    if algo == 1:
        nodes, root = lbvh_single(tris, fe["skeys"], fe["svals"]); out.update(nodes=nodes, leaves=None, root=root, layout=0)
    elif algo == 0:
        nodes, _ = lbvh_two(tris, fe["skeys"], fe["svals"]); out.update(nodes=nodes, leaves=None, root=0, layout=0)
    elif algo == 2:
        nodes, leaves, st = ploc(fe["boxes"], fe["svals"]); out.update(nodes=nodes, leaves=leaves, root=0, layout=1, stats=st)
    elif algo == 3:
        nodes, leaves, st = hploc(fe["boxes"], fe["skeys"], fe["svals"]); out.update(nodes=nodes, leaves=leaves, root=0, layout=1, stats=st)
    else:
        raise ValueError(algo)
    return out


# ---- checks -------------------------------------------------------------------------------------------------------
def sah_bvh2(nodes, leaves, root, n, layout):
    f = C.c_float()
    c = lib().orc_sah_bvh2(nodes.ctypes.data, _p(leaves), root, n, layout, C.byref(f))
    return float(c), float(f.value)


def validate_bvh2(nodes, leaves, root, n, layout) -> int:
    return int(lib().orc_validate_bvh2(nodes.ctypes.data, _p(leaves), root, n, layout))


def topology_hash(nodes, leaves, root, n, layout) -> int:
    return int(lib().orc_topology_hash(nodes.ctypes.data, _p(leaves), root, n, layout))


def fnv1a(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    return int(lib().orc_fnv1a(a.ctypes.data, a.nbytes))


def ploc_to_lbvh_layout(nodes, leaves):
    n = leaves.shape[0]
    out = np.zeros(2 * n - 1, dtype=BVH2_NODE)
    lib().orc_ploc_to_lbvh_layout(nodes.ctypes.data, leaves.ctypes.data, n, out.ctypes.data)
    return out


def binned_sah_build(tris):
    n = tris.shape[0]
    nodes = np.zeros(3 * n - 1, dtype=SAH_NODE)
    total = lib().orc_binned_sah_build(tris.ctypes.data, n, nodes.ctypes.data)
    return nodes, int(total)


def sah_binned(nodes, total, n):
    f = C.c_float()
    c = lib().orc_sah_binned(nodes.ctypes.data, total, n, C.byref(f))
    return float(c), float(f.value)


def collapse4(nodes, leaves, root, n, layout):
    w = np.zeros(n, dtype=BVH4_NODE); pn = np.zeros(n, dtype=PRIM_NODE)
    total = lib().orc_collapse4(nodes.ctypes.data, _p(leaves), root, n, layout, w.ctypes.data, pn.ctypes.data)
    return w, pn, int(total)


RAY = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("tmin", "<f4"), ("tmax", "<f4")])


def generate_rays(camera: np.ndarray, width: int, height: int) -> np.ndarray:
    rays = np.zeros(width * height, dtype=RAY)
    lib().orc_generate_rays(np.ascontiguousarray(camera).ctypes.data, rays.ctypes.data, width, height)
    return rays


def trace_while(rays, tris, nodes_lbvh, transform, root, width, n_internal):
    """-> (rgba uint8[width*width*4], rays whose stack exceeded the reference's 32 entries)"""
    rgba = np.zeros(width * width * 4, dtype=np.uint8); ov = C.c_uint32()
    lib().orc_trace_while(np.ascontiguousarray(rays).ctypes.data, tris.ctypes.data, np.ascontiguousarray(nodes_lbvh).ctypes.data,
                          np.ascontiguousarray(transform).ctypes.data, rgba.ctypes.data, root, width, width, n_internal, C.byref(ov))
    return rgba, int(ov.value)


def topology_hash4(w, pn, total, n) -> int:
    return int(lib().orc_topology_hash4(np.ascontiguousarray(w).ctypes.data, np.ascontiguousarray(pn).ctypes.data, total, n))


def sah_bvh4(w, pn, prim_boxes, total, n):
    f = C.c_float()
    c = lib().orc_sah_bvh4(w.ctypes.data, pn.ctypes.data, prim_boxes.ctypes.data, total, n, C.byref(f))
    return float(c), float(f.value)


# ---- the reference's own Utility.cpp (oracle/_ref/libref_utility.so), when built ------------------------------------
_ref = None


def ref_utility():
    """The reference's unmodified src/Utility.cpp behind extern "C" shims, or None if oracle/_ref was not built."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_UTILITY):
            return None
        L = C.CDLL(REF_UTILITY)
        vp, u32 = C.c_void_p, C.c_uint32
        L.ref_calculateLbvhCost.argtypes = [vp, u32, u32, u32]; L.ref_calculateLbvhCost.restype = C.c_float
        for name in ("ref_checkLbvhRootAabb", "ref_checkLBvhCorrectness"):
            getattr(L, name).argtypes = [vp, u32, u32, u32]; getattr(L, name).restype = C.c_int
        L.ref_checkPlocBvh2Correctness.argtypes = [vp, vp, u32, u32, u32]; L.ref_checkPlocBvh2Correctness.restype = C.c_int
        L.ref_checkLBvh4Correctness.argtypes = [vp, vp, u32, u32]; L.ref_checkLBvh4Correctness.restype = C.c_int
        L.ref_calculatebvh4Cost.argtypes = [vp, vp, vp, u32, u32, u32]; L.ref_calculatebvh4Cost.restype = C.c_float
        L.ref_calculateBinnedSahBvhCost.argtypes = [vp, u32, u32]; L.ref_calculateBinnedSahBvhCost.restype = C.c_float
        L.ref_loadScene.argtypes = [C.c_char_p, C.c_char_p, vp, u32]; L.ref_loadScene.restype = u32
        L.ref_primRefs.argtypes = [vp, u32, vp]; L.ref_primRefs.restype = u32
        L.ref_sizeof.argtypes = [C.c_int]; L.ref_sizeof.restype = u32
        _ref = L
    return _ref


# ---- the reference's PLOC++ kernels under the CPU SIMT emulator (tools/oracle/ref_emulator.cpp -> oracle/_ref/libref_ploc_emu.so, libref_lbvh_emu.so) ----
REF_PLOC_EMU = os.path.join(_HERE, "_ref", "libref_ploc_emu.so")
_emu = None


def ref_emu_ploc(boxes, svals):
    """SetupClusters + Ploc / SinglePassPloc of the reference (src/Ploc++Kernel.h) driven by its host loop (src/PLOC++Bvh.cpp:82-152),
    executed on the CPU.  -> (nodes, leaves, iterations) or None when the library is not built (needs /root/reference)."""
    global _emu
    if not os.path.exists(REF_PLOC_EMU):
        return None
    if _emu is None:
        _emu = C.CDLL(REF_PLOC_EMU)
        _emu.ref_emu_ploc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    n = boxes.shape[0]
    boxes = np.ascontiguousarray(boxes); svals = np.ascontiguousarray(svals, dtype=np.uint32)
    nodes = np.zeros(n - 1, dtype=BVH2_NODE); leaves = np.zeros(n, dtype=PRIMREF); it = C.c_uint32()
    rc = _emu.ref_emu_ploc(boxes.ctypes.data, svals.ctypes.data, n, nodes.ctypes.data, leaves.ctypes.data, C.byref(it))
    if rc != 0:
        raise RuntimeError(f"ref_emu_ploc failed: {rc}")
    return nodes, leaves, int(it.value)


REF_HPLOC_EMU = os.path.join(_HERE, "_ref", "libref_hploc_emu.so")
_emu_h = None


def ref_emu_hploc(boxes, skeys, svals, cover_all=False):
    """SetupClusters + HPloc of the reference (src/HplocKernel.h, host src/Hploc.cpp:83-121) executed on the CPU by the same SIMT emulator that runs the
    Ploc / collapse kernels.  The same header runs unmodified on the MI355X (ref_hploc): comparing the two checks the EMULATOR against silicon.
    -> (nodes, leaves, merged) or None when the library is not built (needs /root/reference)."""
    global _emu_h
    if not os.path.exists(REF_HPLOC_EMU):
        return None
    if _emu_h is None:
        _emu_h = C.CDLL(REF_HPLOC_EMU)
        _emu_h.ref_emu_hploc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
    n = boxes.shape[0]
    boxes = np.ascontiguousarray(boxes); skeys = np.ascontiguousarray(skeys, dtype=np.uint32); svals = np.ascontiguousarray(svals, dtype=np.uint32)
    nodes = np.zeros(max(n - 1, 1), dtype=BVH2_NODE); leaves = np.zeros(n, dtype=PRIMREF); merged = C.c_uint32()
    rc = _emu_h.ref_emu_hploc(boxes.ctypes.data, skeys.ctypes.data, svals.ctypes.data, n, nodes.ctypes.data, leaves.ctypes.data, C.byref(merged), int(cover_all))
    if rc != 0:
        raise RuntimeError(f"ref_emu_hploc failed: {rc}")
    return nodes[: n - 1], leaves, int(merged.value)


REF_LBVH_EMU = os.path.join(_HERE, "_ref", "libref_lbvh_emu.so")
_emu_l = None


def ref_emu_collapse(nodes, leaves, root, n, layout):
    """the reference's CollapseToWide4Bvh kernel (LBVH layout: src/TwoPassLbvhKernel.h:237-336; PLOC layout: src/Ploc++Kernel.h:364-465) with its
    host set-up, executed on the CPU with the whole grid resident.  -> (Bvh4Node[n_wide], PrimNode[n], n_wide) or None when not built."""
    global _emu, _emu_l
    path = REF_PLOC_EMU if layout == 1 else REF_LBVH_EMU
    if not os.path.exists(path):
        return None
    wide = np.zeros(n, dtype=BVH4_NODE); prims = np.zeros(n, dtype=PRIM_NODE); nw = C.c_uint32()
    if layout == 1:
        if _emu is None:
            _emu = C.CDLL(REF_PLOC_EMU)
            _emu.ref_emu_ploc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        _emu.ref_emu_collapse_ploc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        padded = np.zeros(2 * n, dtype=BVH2_NODE); padded[: n - 1] = nodes          # the kernel reads node[leaf index].m_aabb (values unused)
        lv = np.ascontiguousarray(leaves)
        rc = _emu.ref_emu_collapse_ploc(padded.ctypes.data, lv.ctypes.data, root, n, wide.ctypes.data, prims.ctypes.data, C.byref(nw))
    else:
        if _emu_l is None:
            _emu_l = C.CDLL(REF_LBVH_EMU)
            _emu_l.ref_emu_collapse_lbvh.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        nd = np.ascontiguousarray(nodes)
        rc = _emu_l.ref_emu_collapse_lbvh(nd.ctypes.data, root, n, wide.ctypes.data, prims.ctypes.data, C.byref(nw))
    if rc != 0:
        raise RuntimeError(f"ref_emu_collapse failed: {rc}")
    return wide[: nw.value].copy(), prims, int(nw.value)


# ---- the reference's own device kernels on the GPU (oracle/_ref/*.co driven by oracle/ref_driver.cpp), when built ------
_drv = {}


def ref_driver(nofma: bool = False):
    """Loads oracle/_ref/libref_driver.so and the reference's compiled kernels; None when not built / no GPU."""
    key = bool(nofma)
    if key in _drv:
        return _drv[key]
    if not os.path.exists(REF_DRIVER):
        return None
    L = C.CDLL(REF_DRIVER)
    vp, u32 = C.c_void_p, C.c_uint32
    L.refdrv_error.restype = C.c_char_p
    L.refdrv_init.argtypes = [C.c_char_p, C.c_int]
    L.refdrv_morton.argtypes = [vp, u32, vp, vp, vp]
    L.refdrv_lbvh_single.argtypes = [vp, u32, vp, vp, vp, C.POINTER(u32)]
    L.refdrv_lbvh_two.argtypes = [vp, u32, vp, vp, vp]
    L.refdrv_hploc.argtypes = [vp, u32, vp, vp, vp, vp, C.POINTER(u32), C.c_int]
    L.refdrv_generate_rays.argtypes = [vp, vp, u32, u32]
    L.refdrv_trace_while.argtypes = [vp, vp, u32, vp, u32, vp, vp, u32, u32, u32, u32]
    L.refdrv_trace_kind.argtypes = [C.c_int, vp, vp, u32, vp, u32, vp, vp, vp, u32, u32, u32, u32]
    L.refdrv_extents.argtypes = [vp, u32, vp, vp]
    L.refdrv_primref_frontend.argtypes = [vp, u32, vp, vp, vp]
    L.refdrv_ploc.argtypes = [vp, u32, vp, vp, vp, C.POINTER(u32), C.c_int]
    L.refdrv_collapse.argtypes = [C.c_int, vp, vp, u32, u32, vp, vp, C.POINTER(u32), C.POINTER(u32)]
    rc = L.refdrv_init(os.path.join(_HERE, "_ref").encode(), int(nofma))
    if rc != 0:
        raise RuntimeError("refdrv_init: " + L.refdrv_error().decode())
    _drv[key] = L
    return L


def _rc(L, rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {L.refdrv_error().decode()}")


def _reinit(L, nofma):
    # one process-wide set of loaded modules: re-load when switching flavour
    rc = L.refdrv_init(os.path.join(_HERE, "_ref").encode(), int(nofma))
    if rc != 0:
        raise RuntimeError("refdrv_init: " + L.refdrv_error().decode())


def ref_morton(boxes, scene, nofma=False):
    L = ref_driver(nofma); _reinit(L, nofma); n = boxes.shape[0]
    keys = np.empty(n, dtype=np.uint32); vals = np.empty(n, dtype=np.uint32)
    _rc(L, L.refdrv_morton(boxes.ctypes.data, n, scene.ctypes.data, keys.ctypes.data, vals.ctypes.data), "refdrv_morton")
    return keys, vals


def ref_lbvh_single(tris, skeys, svals, nofma=False):
    L = ref_driver(nofma); _reinit(L, nofma); n = tris.shape[0]
    nodes = np.zeros(2 * n - 1, dtype=BVH2_NODE); root = C.c_uint32()
    _rc(L, L.refdrv_lbvh_single(tris.ctypes.data, n, skeys.ctypes.data, svals.ctypes.data, nodes.ctypes.data, C.byref(root)), "refdrv_lbvh_single")
    return nodes, int(root.value)


def ref_lbvh_two(tris, skeys, svals, nofma=False):
    L = ref_driver(nofma); _reinit(L, nofma); n = tris.shape[0]
    refs = primrefs(tris)
    nodes = np.zeros(2 * n - 1, dtype=BVH2_NODE)
    _rc(L, L.refdrv_lbvh_two(refs.ctypes.data, n, skeys.ctypes.data, svals.ctypes.data, nodes.ctypes.data), "refdrv_lbvh_two")
    return nodes


def ref_hploc(boxes, skeys, svals, nofma=False, cover_all=False):
    L = ref_driver(nofma); _reinit(L, nofma); n = boxes.shape[0]
    nodes = np.zeros(max(n - 1, 1), dtype=BVH2_NODE); leaves = np.zeros(n, dtype=PRIMREF); merged = C.c_uint32()
    _rc(L, L.refdrv_hploc(boxes.ctypes.data, n, skeys.ctypes.data, svals.ctypes.data, nodes.ctypes.data, leaves.ctypes.data, C.byref(merged), int(cover_all)), "refdrv_hploc")
    return nodes[: n - 1], leaves, int(merged.value)


def ref_extents(tris, nofma=False):
    """the reference's CalculateSceneExtents (src/CommonBlocksKernel.h:92-114), its own wave64 flavour, on the GPU -> (Aabb[n], scene Aabb)"""
    L = ref_driver(nofma); _reinit(L, nofma); n = tris.shape[0]
    boxes = np.zeros(n, dtype=AABB); scene = np.zeros(1, dtype=AABB)
    _rc(L, L.refdrv_extents(tris.ctypes.data, n, boxes.ctypes.data, scene.ctypes.data), "refdrv_extents")
    return boxes, scene


def ref_primref_frontend(primrefs, nofma=False):
    """the reference's CalculatePrimRefExtents (src/CommonBlocksKernel.h:116-137, wave64 flavour) + CalculateMortonCodesPrimRef (:387-398) on the GPU, in the host order of
    src/TwoPassLbvh.cpp:40-68 -> (scene Aabb, keys, values)"""
    L = ref_driver(nofma); _reinit(L, nofma); n = primrefs.shape[0]
    primrefs = np.ascontiguousarray(primrefs)
    scene = np.zeros(1, dtype=AABB); keys = np.zeros(n, dtype=np.uint32); vals = np.zeros(n, dtype=np.uint32)
    _rc(L, L.refdrv_primref_frontend(primrefs.ctypes.data, n, scene.ctypes.data, keys.ctypes.data, vals.ctypes.data), "refdrv_primref_frontend")
    return scene, keys, vals


def ref_ploc(boxes, svals, nofma=False, never_single_pass=False):
    """the reference's SetupClusters + Ploc (+ SinglePassPloc below 1024 clusters) kernels, wave64 flavour, on the GPU under the host loop of
    src/PLOC++Bvh.cpp:82-152 -> (nodes, leaves, iterations)"""
    L = ref_driver(nofma); _reinit(L, nofma); n = boxes.shape[0]
    boxes = np.ascontiguousarray(boxes); svals = np.ascontiguousarray(svals, dtype=np.uint32)
    nodes = np.zeros(n - 1, dtype=BVH2_NODE); leaves = np.zeros(n, dtype=PRIMREF); it = C.c_uint32()
    _rc(L, L.refdrv_ploc(boxes.ctypes.data, n, svals.ctypes.data, nodes.ctypes.data, leaves.ctypes.data, C.byref(it), int(never_single_pass)), "refdrv_ploc")
    return nodes, leaves, int(it.value)


def ref_collapse(nodes, leaves, root, n, layout, nofma=False):
    """the reference's CollapseToWide4Bvh kernel (layout 0: src/TwoPassLbvhKernel.h:237-336, layout 1: src/Ploc++Kernel.h:364-465) on the GPU with
    the reference's host set-up -> (Bvh4Node[n_wide], PrimNode[n], n_wide, leaves placed)"""
    L = ref_driver(nofma); _reinit(L, nofma)
    wide = np.zeros(2 * n, dtype=BVH4_NODE); prims = np.zeros(n, dtype=PRIM_NODE); nw = C.c_uint32(); cnt = C.c_uint32()
    nd = np.ascontiguousarray(nodes); lv = np.ascontiguousarray(leaves) if leaves is not None else None
    _rc(L, L.refdrv_collapse(layout, nd.ctypes.data, lv.ctypes.data if lv is not None else None, root, n, wide.ctypes.data, prims.ctypes.data,
                             C.byref(nw), C.byref(cnt)), "refdrv_collapse")
    return wide[: nw.value].copy(), prims, int(nw.value), int(cnt.value)


def ref_generate_rays(camera, width, height, nofma=False):
    L = ref_driver(nofma); _reinit(L, nofma)
    rays = np.zeros(width * height, dtype=RAY)
    _rc(L, L.refdrv_generate_rays(np.ascontiguousarray(camera).ctypes.data, rays.ctypes.data, width, height), "refdrv_generate_rays")
    return rays


def ref_trace_kind(kind, rays, tris, nodes_lbvh, transform, root, width, n_internal, nofma=False):
    """the reference's BvhTraversalRestartTrail (1) / BvhTraversalifif (2) / BvhTraversalSpeculativeWhile (3) -> (rgba, tests per ray)"""
    L = ref_driver(nofma); _reinit(L, nofma)
    rgba = np.zeros(width * width * 4, dtype=np.uint8); cnt = np.zeros(width * width, dtype=np.uint32)
    nodes_lbvh = np.ascontiguousarray(nodes_lbvh)
    _rc(L, L.refdrv_trace_kind(kind, np.ascontiguousarray(rays).ctypes.data, tris.ctypes.data, tris.shape[0], nodes_lbvh.ctypes.data, nodes_lbvh.shape[0],
                               np.ascontiguousarray(transform).ctypes.data, rgba.ctypes.data, cnt.ctypes.data, root, width, width, n_internal), "refdrv_trace_kind")
    return rgba, cnt


def ref_trace_while(rays, tris, nodes_lbvh, transform, root, width, n_internal, nofma=False):
    L = ref_driver(nofma); _reinit(L, nofma)
    rgba = np.zeros(width * width * 4, dtype=np.uint8)
    nodes_lbvh = np.ascontiguousarray(nodes_lbvh)
    _rc(L, L.refdrv_trace_while(np.ascontiguousarray(rays).ctypes.data, tris.ctypes.data, tris.shape[0], nodes_lbvh.ctypes.data, nodes_lbvh.shape[0],
                                np.ascontiguousarray(transform).ctypes.data, rgba.ctypes.data, root, width, width, n_internal), "refdrv_trace_while")
    return rgba
