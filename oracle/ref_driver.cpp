// ref_driver.cpp — host driver that runs the REFERENCE's own device kernels on the MI355X.
//
// TEST INFRASTRUCTURE ONLY.  The kernels are the reference's src/*Kernel.h, compiled unmodified by hipcc into
// oracle/_ref/<Header>.co (oracle/Makefile; binaries only, never sources).  This file is own code: it loads those code
// objects with hipModuleLoad and launches the kernels in the order and with the buffers the reference's host code uses
// (src/SinglePassLbvh.cpp:102-131, src/TwoPassLbvh.cpp:99-143, src/Hploc.cpp:83-121, src/PLOC++Bvh.cpp:45-57) — the
// reference's own host layer needs Orochi, which is not in its tree.
//
// What runs on gfx950: every kernel without wave-size assumptions (Morton, both LBVH builders, SetupClusters, both collapse
// kernels) and HPloc, whose 32-thread workgroups occupy the lower half of a wave64, from the plain builds; CalculateSceneExtents,
// Ploc and SinglePassPloc — which reduce / scan across a wave with the reference's `WarpSize` constant — from the reference's own
// WAVE64 flavour of the same unmodified headers (*.w64.co: src/Common.h:100-106 selects WarpSize = 64 under -D__gfx90a__=1,
// oracle/Makefile).  Round 5; rounds 1-4 wrongly held that these kernels could not run on wave64 hardware.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
using u32 = uint32_t;
struct Mod { hipModule_t m = nullptr; };
Mod g_common, g_single, g_two, g_hploc, g_trav, g_common64, g_ploc64;
std::string g_err;

#define RT(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); return -(int)e_; } } while (0)

int load(Mod& md, const std::string& path) {
    if (md.m) { (void)hipModuleUnload(md.m); md.m = nullptr; }
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { g_err = "cannot open " + path; return -1; }
    std::fseek(f, 0, SEEK_END); long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<char> buf((size_t)sz);
    size_t rd = std::fread(buf.data(), 1, (size_t)sz, f); std::fclose(f);
    if (rd != (size_t)sz) { g_err = "short read " + path; return -1; }
    RT(hipModuleLoadData(&md.m, buf.data()));
    return 0;
}

int launch(hipModule_t m, const char* name, u32 threads, u32 block, void** args) {
    hipFunction_t fn;
    RT(hipModuleGetFunction(&fn, m, name));
    const u32 grid = (threads + block - 1) / block;   // Kernel::launch(nx[, tx]) — src/Kernel.cpp:155-168
    RT(hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, nullptr, args, nullptr));
    return 0;
}

template <typename T> struct Dev {
    T* p = nullptr; size_t n = 0;
    int alloc(size_t count, int fill = -2) { n = count; RT(hipMalloc(&p, count * sizeof(T) + 64)); if (fill != -2) RT(hipMemset(p, fill, count * sizeof(T))); return 0; }
    int up(const void* h) { RT(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice)); return 0; }
    int down(void* h) { RT(hipMemcpy(h, p, n * sizeof(T), hipMemcpyDeviceToHost)); return 0; }
    ~Dev() { if (p) (void)hipFree(p); }
};
struct B24 { char b[24]; }; struct B28 { char b[28]; }; struct B32 { char b[32]; }; struct B64 { char b[64]; }; struct U2 { u32 x, y; };
#define TRY(x) do { int r_ = (x); if (r_) return r_; } while (0)
} // namespace

extern "C" {

const char* refdrv_error() { return g_err.c_str(); }

// dir: oracle/_ref ; nofma != 0 selects the *.nofma.co builds (FP contraction off)
int refdrv_init(const char* dir, int nofma) {
    const std::string d(dir), sfx = nofma ? ".nofma.co" : ".co";
    TRY(load(g_common, d + "/CommonBlocksKernel" + sfx));
    TRY(load(g_single, d + "/SinglePassLbvhKernel" + sfx));
    TRY(load(g_two, d + "/TwoPassLbvhKernel" + sfx));
    TRY(load(g_hploc, d + "/HplocKernel" + sfx));
    TRY(load(g_trav, d + "/TraversalKernel" + sfx));
    TRY(load(g_common64, d + "/CommonBlocksKernel.w64" + sfx));
    TRY(load(g_ploc64, d + "/Ploc++Kernel.w64" + sfx));
    return 0;
}

// CalculateMortonCodes (src/CommonBlocksKernel.h:374-385); host: src/PLOC++Bvh.cpp:45-57
int refdrv_morton(const void* h_boxes, u32 n, const void* h_scene, u32* h_keys, u32* h_vals) {
    Dev<B24> boxes, scene; Dev<u32> keys, vals;
    TRY(boxes.alloc(n)); TRY(boxes.up(h_boxes)); TRY(scene.alloc(1)); TRY(scene.up(h_scene)); TRY(keys.alloc(n, 0)); TRY(vals.alloc(n, 0));
    void* args[] = { &boxes.p, &scene.p, &keys.p, &vals.p, &n };
    TRY(launch(g_common.m, "CalculateMortonCodes", n, 256, args));
    RT(hipDeviceSynchronize());
    TRY(keys.down(h_keys)); TRY(vals.down(h_vals));
    return 0;
}

// InitBvhNodes + BvhBuildAndFit (src/SinglePassLbvhKernel.h:27-126); host: src/SinglePassLbvh.cpp:102-131
int refdrv_lbvh_single(const void* h_tris, u32 n, const u32* h_skeys, const u32* h_svals, void* h_nodes, u32* root) {
    const u32 ni = n - 1;
    Dev<B64> tris; Dev<B32> nodes; Dev<u32> skeys, svals; Dev<U2> spans; Dev<int> counter;
    TRY(tris.alloc(n)); TRY(tris.up(h_tris)); TRY(nodes.alloc(2 * (size_t)n - 1, 0)); TRY(skeys.alloc(n)); TRY(skeys.up(h_skeys));
    TRY(svals.alloc(n)); TRY(svals.up(h_svals)); TRY(spans.alloc(n, 0)); TRY(counter.alloc(n, 0));
    u32 nleaf = n, nint = ni;
    { void* a[] = { &tris.p, &nodes.p, &svals.p, &nint, &nleaf }; TRY(launch(g_single.m, "InitBvhNodes", n, 256, a)); }
    { void* a[] = { &nodes.p, &counter.p, &spans.p, &skeys.p, &nleaf, &nint }; TRY(launch(g_single.m, "BvhBuildAndFit", n, 256, a)); }
    RT(hipDeviceSynchronize());
    TRY(nodes.down(h_nodes));
    std::vector<int> c(n); TRY(counter.down(c.data()));
    *root = (u32)c[n - 1];                                   // src/SinglePassLbvh.cpp:131
    return 0;
}

// InitBvhNodesPrimRef + BvhBuild + FitBvhNodes (src/TwoPassLbvhKernel.h:164-235); host: src/TwoPassLbvh.cpp:99-143
int refdrv_lbvh_two(const void* h_primrefs, u32 n, const u32* h_skeys, const u32* h_svals, void* h_nodes) {
    const u32 ni = n - 1; const size_t total = 2 * (size_t)n - 1;
    Dev<B28> refs; Dev<B32> nodes; Dev<u32> skeys, svals, parent, flags;
    TRY(refs.alloc(n)); TRY(refs.up(h_primrefs)); TRY(nodes.alloc(total, 0)); TRY(skeys.alloc(n)); TRY(skeys.up(h_skeys));
    TRY(svals.alloc(n)); TRY(svals.up(h_svals)); TRY(parent.alloc(total, 0)); TRY(flags.alloc(total, 0));
    u32 nleaf = n, nint = ni;
    { void* a[] = { &refs.p, &nodes.p, &parent.p, &svals.p, &nleaf, &nint }; TRY(launch(g_two.m, "InitBvhNodesPrimRef", n, 256, a)); }
    { void* a[] = { &nodes.p, &parent.p, &skeys.p, &nleaf, &nint }; TRY(launch(g_two.m, "BvhBuild", ni, 256, a)); }
    { void* a[] = { &nodes.p, &parent.p, &flags.p, &nleaf, &nint }; TRY(launch(g_two.m, "FitBvhNodes", n, 256, a)); }
    RT(hipDeviceSynchronize());
    TRY(nodes.down(h_nodes));
    return 0;
}

// SetupClusters + HPloc (src/HplocKernel.h:39-56,257-315); host: src/Hploc.cpp:83-121.  cover_all != 0 launches ceil(n/32)
// workgroups instead of the reference's ceil((n-1)/32) (which leaves leaf n-1 without a thread when (n-1)%32 == 0, App. B).
int refdrv_hploc(const void* h_boxes, u32 n, const u32* h_skeys, const u32* h_svals, void* h_nodes, void* h_leaves, u32* merged, int cover_all) {
    const u32 ni = n - 1;
    Dev<B24> boxes; Dev<B32> nodes; Dev<B28> leaves; Dev<u32> skeys, svals, idx, parent, cnt;
    TRY(boxes.alloc(n)); TRY(boxes.up(h_boxes)); TRY(nodes.alloc(ni ? ni : 1, 0)); TRY(leaves.alloc(n, 0)); TRY(skeys.alloc(n)); TRY(skeys.up(h_skeys));
    TRY(svals.alloc(n)); TRY(svals.up(h_svals)); TRY(idx.alloc(n, 0xFF)); TRY(parent.alloc(n, 0xFF)); TRY(cnt.alloc(1, 0));
    u32 prim = n, ncl = n, nint = ni;
    { void* a[] = { &nodes.p, &leaves.p, &svals.p, &boxes.p, &idx.p, &parent.p, &prim }; TRY(launch(g_hploc.m, "SetupClusters", n, 256, a)); }
    { void* a[] = { &nodes.p, &leaves.p, &skeys.p, &idx.p, &parent.p, &cnt.p, &ncl, &nint }; TRY(launch(g_hploc.m, "HPloc", cover_all ? n : ni, 32, a)); }
    RT(hipDeviceSynchronize());
    TRY(nodes.down(h_nodes)); TRY(leaves.down(h_leaves)); TRY(cnt.down(merged));
    return 0;
}

// CalculateSceneExtents (src/CommonBlocksKernel.h:92-114, wave64 flavour); host: src/PLOC++Bvh.cpp:19-37 (extent reset to +-FltMax, launch over
// primitiveCount threads in ReductionBlockSize = 256 workgroups)
int refdrv_extents(const void* h_tris, u32 n, void* h_boxes, void* h_scene) {
    Dev<B64> tris; Dev<B24> boxes, scene;
    TRY(tris.alloc(n)); TRY(tris.up(h_tris)); TRY(boxes.alloc(n, 0)); TRY(scene.alloc(1));
    const float fmax = 3.402823466e+38f; const float ext[6] = { fmax, fmax, fmax, -fmax, -fmax, -fmax };   // Aabb::reset(), src/Common.h:327-331
    TRY(scene.up(ext));
    void* a[] = { &tris.p, &boxes.p, &scene.p, &n };
    TRY(launch(g_common64.m, "CalculateSceneExtents", n, 256, a));
    RT(hipDeviceSynchronize());
    TRY(boxes.down(h_boxes)); TRY(scene.down(h_scene));
    return 0;
}

// The PrimRef flavours of the LBVH front end: CalculatePrimRefExtents (src/CommonBlocksKernel.h:116-137, wave64 flavour) + CalculateMortonCodesPrimRef (:387-398); host:
// src/TwoPassLbvh.cpp:40-68, src/SinglePassLbvh.cpp:41-69 (extent reset to +-FltMax, both kernels launched over primitiveCount threads).  h_primrefs: PrimRef[n] as
// Utility::doEarlySplitClipping leaves them when nothing is split (src/Utility.cpp:456-477: {i, Aabb grown over the three vertices}).
int refdrv_primref_frontend(const void* h_primrefs, u32 n, void* h_scene, u32* h_keys, u32* h_vals) {
    Dev<B28> refs; Dev<B24> scene; Dev<u32> keys, vals;
    TRY(refs.alloc(n)); TRY(refs.up(h_primrefs)); TRY(scene.alloc(1)); TRY(keys.alloc(n, 0)); TRY(vals.alloc(n, 0));
    const float fmax = 3.402823466e+38f; const float ext[6] = { fmax, fmax, fmax, -fmax, -fmax, -fmax };   // Aabb::reset(), src/Common.h:327-331
    TRY(scene.up(ext));
    { void* a[] = { &refs.p, &scene.p, &n }; TRY(launch(g_common64.m, "CalculatePrimRefExtents", n, 256, a)); }
    { void* a[] = { &refs.p, &scene.p, &keys.p, &vals.p, &n }; TRY(launch(g_common.m, "CalculateMortonCodesPrimRef", n, 256, a)); }
    RT(hipDeviceSynchronize());
    TRY(scene.down(h_scene)); TRY(keys.down(h_keys)); TRY(vals.down(h_vals));
    return 0;
}

// SetupClusters + the host loop of Ploc / SinglePassPloc (src/Ploc++Kernel.h:39-55,98-362, wave64 flavour); host: src/PLOC++Bvh.cpp:82-152 —
// the three counters are cleared and the merged count read back (D2H) per iteration, the index buffers swap, below PlocBlockSize clusters one
// SinglePassPloc launch finishes.  *iterations: passes of the loop.  never_single_pass: 0 = the reference's rule (SinglePassPloc below 1024
// clusters); 1 keeps `Ploc` iterating down to one cluster, which separates the two kernels' behaviour on silicon (a test's instrument).
int refdrv_ploc(const void* h_boxes, u32 n, const u32* h_svals, void* h_nodes, void* h_leaves, u32* iterations, int never_single_pass) {
    if (n < 2) { g_err = "n < 2"; return -1; }
    const u32 ni = n - 1;
    Dev<B24> boxes; Dev<B32> nodes; Dev<B28> leaves; Dev<u32> svals; Dev<int> idx0, idx1, merged, offsum, counter;
    TRY(boxes.alloc(n)); TRY(boxes.up(h_boxes)); TRY(nodes.alloc(ni, 0)); TRY(leaves.alloc(n, 0)); TRY(svals.alloc(n)); TRY(svals.up(h_svals));
    TRY(idx0.alloc(n, 0xFF)); TRY(idx1.alloc(n, 0xFF)); TRY(merged.alloc(1, 0)); TRY(offsum.alloc(1, 0)); TRY(counter.alloc(1, 0));
    u32 prim = n, nint = ni;
    { void* a[] = { &nodes.p, &leaves.p, &svals.p, &boxes.p, &idx0.p, &prim }; TRY(launch(g_ploc64.m, "SetupClusters", n, 256, a)); }
    bool swap = false; u32 c = n, iters = 0;
    while (c > 1) {
        RT(hipMemset(merged.p, 0, 4)); RT(hipMemset(offsum.p, 0, 4)); RT(hipMemset(counter.p, 0, 4));
        int* i0 = !swap ? idx0.p : idx1.p; int* i1 = !swap ? idx1.p : idx0.p;
        ++iters;
        if (c < 1024u && !never_single_pass) {                         // PlocBlockSize, src/Common.h:593
            void* a[] = { &i0, &nodes.p, &leaves.p, &c, &nint };
            TRY(launch(g_ploc64.m, "SinglePassPloc", c, 1024, a));
            break;
        }
        void* a[] = { &i0, &i1, &nodes.p, &leaves.p, &merged.p, &offsum.p, &counter.p, &c, &nint };
        TRY(launch(g_ploc64.m, "Ploc", c, 1024, a));
        int m = 0; RT(hipMemcpy(&m, merged.p, 4, hipMemcpyDeviceToHost));
        if (m <= 0) { g_err = "Ploc merged nothing"; return -2; }
        c -= (u32)m; swap = !swap;
        if (iters > 100000) { g_err = "Ploc does not terminate"; return -3; }
    }
    RT(hipDeviceSynchronize());
    TRY(nodes.down(h_nodes)); TRY(leaves.down(h_leaves));
    if (iterations) *iterations = iters;
    return 0;
}

// CollapseToWide4Bvh — layout 0: src/TwoPassLbvhKernel.h:237-336 (host src/TwoPassLbvh.cpp:154-183); layout 1: src/Ploc++Kernel.h:364-465 (host
// src/PLOC++Bvh.cpp:154-184).  Task queue all-invalid except the root task, internal-node offset 1, ceil(2 n / 3) threads, all of which must be
// RESIDENT (the kernel spins until its task appears): the caller keeps n below what the device holds (256-thread workgroups here).
// h_nodes: layout 0 Bvh2Node[2n-1]; layout 1 Bvh2Node[n-1] (uploaded into a zeroed 2n array: the kernel reads bvh2Nodes[leaf index].m_aabb, :395-396,
// values unused).  h_wide: Bvh4Node[2n]; h_prims: PrimNode[n]; *n_wide = the internal-node offset after the launch.
int refdrv_collapse(int layout, const void* h_nodes, const void* h_leaves, u32 root, u32 n, void* h_wide, void* h_prims, u32* n_wide, u32* task_count) {
    if (n < 2) { g_err = "n < 2"; return -1; }
    struct B128 { char b[128]; }; struct B8 { char b[8]; };
    const u32 ni = n - 1;
    Dev<B32> nodes; Dev<B28> leaves; Dev<B128> wide; Dev<B8> prims; Dev<U2> taskq; Dev<u32> count, offset;
    const size_t nn = layout == 1 ? 2 * (size_t)n : 2 * (size_t)n - 1;
    TRY(nodes.alloc(nn, 0)); RT(hipMemcpy(nodes.p, h_nodes, (layout == 1 ? (size_t)ni : nn) * 32, hipMemcpyHostToDevice));
    if (layout == 1) { TRY(leaves.alloc(n)); TRY(leaves.up(h_leaves)); }
    TRY(wide.alloc(2 * (size_t)n)); TRY(prims.alloc(n));
    {   // GpuMemory::reset() is a memset 0 (the structs' INVALID defaults are NOT what the device sees) — src/TwoPassLbvh.cpp:154-155
        RT(hipMemset(wide.p, 0, 2 * (size_t)n * 128)); RT(hipMemset(prims.p, 0, (size_t)n * 8));
    }
    std::vector<U2> q(n, U2{0xFFFFFFFFu, 0xFFFFFFFFu}); q[0] = U2{root, 0xFFFFFFFFu};
    TRY(taskq.alloc(n)); TRY(taskq.up(q.data()));
    TRY(count.alloc(1, 0)); TRY(offset.alloc(1)); const u32 one = 1; TRY(offset.up(&one));
    u32 nint = ni, nleaf = n;
    const u32 threads = (2 * n + 2) / 3;
    {   // refuse a launch that cannot be resident as a whole: it would spin for ever (and hang the device)
        hipFunction_t fn; RT(hipModuleGetFunction(&fn, layout == 1 ? g_ploc64.m : g_two.m, "CollapseToWide4Bvh"));
        int per_cu = 0, dev = 0, cus = 0;
        RT(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0));
        RT(hipGetDevice(&dev)); RT(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if ((size_t)(threads + 255) / 256 > (size_t)per_cu * cus) { g_err = "CollapseToWide4Bvh: grid not resident (" + std::to_string(per_cu) + " x " + std::to_string(cus) + " workgroups)"; return -4; }
    }
    if (layout == 1) {
        void* a[] = { &nodes.p, &leaves.p, &wide.p, &prims.p, &taskq.p, &count.p, &offset.p, &nint, &nleaf };
        TRY(launch(g_ploc64.m, "CollapseToWide4Bvh", threads, 256, a));
    } else {
        void* a[] = { &nodes.p, &wide.p, &prims.p, &taskq.p, &count.p, &offset.p, &nint, &nleaf };
        TRY(launch(g_two.m, "CollapseToWide4Bvh", threads, 256, a));
    }
    RT(hipDeviceSynchronize());
    TRY(wide.down(h_wide)); TRY(prims.down(h_prims));
    u32 off = 0, cnt = 0; RT(hipMemcpy(&off, offset.p, 4, hipMemcpyDeviceToHost)); RT(hipMemcpy(&cnt, count.p, 4, hipMemcpyDeviceToHost));
    if (n_wide) *n_wide = off;
    if (task_count) *task_count = cnt;
    return 0;
}

// GenerateRays (src/CommonBlocksKernel.h:432-463); host: src/TwoPassLbvh.cpp:225-242 (8x8 workgroups)
int refdrv_generate_rays(const void* h_camera, void* h_rays, u32 width, u32 height) {
    Dev<B64> cam; Dev<B32> rays;
    TRY(cam.alloc(1)); TRY(cam.up(h_camera)); TRY(rays.alloc((size_t)width * height, 0));
    hipFunction_t fn; RT(hipModuleGetFunction(&fn, g_common.m, "GenerateRays"));
    void* a[] = { &cam.p, &rays.p, &width, &height };
    RT(hipModuleLaunchKernel(fn, (width + 7) / 8, (height + 7) / 8, 1, 8, 8, 1, 0, nullptr, a, nullptr));
    RT(hipDeviceSynchronize());
    TRY(rays.down(h_rays));
    return 0;
}

// BvhTraversalWhile (src/TraversalKernel.h:238-335); host launch shape as src/TwoPassLbvh.cpp:275-294
int refdrv_trace_while(const void* h_rays, const void* h_tris, u32 n_tris, const void* h_nodes, u32 n_nodes, const void* h_transform,
                       unsigned char* h_rgba, u32 root, u32 width, u32 height, u32 n_internal) {
    Dev<B32> rays, nodes; Dev<B64> tris, xf; Dev<unsigned char> rgba;
    TRY(rays.alloc((size_t)width * height)); TRY(rays.up(h_rays)); TRY(tris.alloc(n_tris)); TRY(tris.up(h_tris));
    TRY(nodes.alloc(n_nodes)); TRY(nodes.up(h_nodes)); TRY(xf.alloc(1)); TRY(xf.up(h_transform)); TRY(rgba.alloc((size_t)width * height * 4, 0));
    hipFunction_t fn; RT(hipModuleGetFunction(&fn, g_trav.m, "BvhTraversalWhile"));
    void* a[] = { &rays.p, &tris.p, &nodes.p, &xf.p, &rgba.p, &root, &width, &height, &n_internal };
    RT(hipModuleLaunchKernel(fn, (width + 7) / 8, (height + 7) / 8, 1, 8, 8, 1, 0, nullptr, a, nullptr));
    RT(hipDeviceSynchronize());
    TRY(rgba.down(h_rgba));
    return 0;
}

// the other three traversal kernels (src/TraversalKernel.h:49-146 restart trail, :148-236 if-if, :337-451 speculative while-while);
// kind as in bvh_trace_kind; h_counter (u32 per ray) is the reference's rayCounter (kinds 1 and 2 only)
int refdrv_trace_kind(int kind, const void* h_rays, const void* h_tris, u32 n_tris, const void* h_nodes, u32 n_nodes, const void* h_transform,
                      unsigned char* h_rgba, u32* h_counter, u32 root, u32 width, u32 height, u32 n_internal) {
    Dev<B32> rays, nodes; Dev<B64> tris, xf; Dev<unsigned char> rgba; Dev<u32> cnt;
    TRY(rays.alloc((size_t)width * height)); TRY(rays.up(h_rays)); TRY(tris.alloc(n_tris)); TRY(tris.up(h_tris));
    TRY(nodes.alloc(n_nodes)); TRY(nodes.up(h_nodes)); TRY(xf.alloc(1)); TRY(xf.up(h_transform)); TRY(rgba.alloc((size_t)width * height * 4, 0));
    TRY(cnt.alloc((size_t)width * height, 0));
    const char* name = kind == 1 ? "BvhTraversalRestartTrail" : kind == 2 ? "BvhTraversalifif" : "BvhTraversalSpeculativeWhile";
    hipFunction_t fn; RT(hipModuleGetFunction(&fn, g_trav.m, name));
    void* with_counter[] = { &rays.p, &cnt.p, &tris.p, &nodes.p, &xf.p, &rgba.p, &root, &width, &height, &n_internal };
    void* without[] = { &rays.p, &tris.p, &nodes.p, &xf.p, &rgba.p, &root, &width, &height, &n_internal };
    RT(hipModuleLaunchKernel(fn, (width + 7) / 8, (height + 7) / 8, 1, 8, 8, 1, 0, nullptr, kind == 3 ? without : with_counter, nullptr));
    RT(hipDeviceSynchronize());
    TRY(rgba.down(h_rgba));
    if (h_counter) TRY(cnt.down(h_counter));
    return 0;
}

} // extern "C"
