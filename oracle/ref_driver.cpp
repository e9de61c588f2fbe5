// ref_driver.cpp — host driver that runs the REFERENCE's own device kernels on the MI355X.
//
// TEST INFRASTRUCTURE ONLY.  The kernels are the reference's src/*Kernel.h, compiled unmodified by hipcc into
// oracle/_ref/<Header>.co (oracle/Makefile; binaries only, never sources).  This file is own code: it loads those code
// objects with hipModuleLoad and launches the kernels in the order and with the buffers the reference's host code uses
// (src/SinglePassLbvh.cpp:102-131, src/TwoPassLbvh.cpp:99-143, src/Hploc.cpp:83-121, src/PLOC++Bvh.cpp:45-57) — the
// reference's own host layer needs Orochi, which is not in its tree.
//
// What can run on gfx950: every kernel without wave-size assumptions (Morton, both LBVH builders, SetupClusters) and
// HPloc, whose 32-thread workgroups occupy the lower half of a wave64.  CalculateSceneExtents and Ploc hard-code
// WarpSize = 32 for cross-lane reductions over larger workgroups (src/Common.h:100-106) and are wrong on wave64 hardware;
// they are not driven here.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
using u32 = uint32_t;
struct Mod { hipModule_t m = nullptr; };
Mod g_common, g_single, g_two, g_hploc, g_trav;
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
