// builders.hpp — C++ host mirror of the reference's builder API over the C ABI (include/bvh_mi355x.h).
//
// Same namespace, class names, method signatures and public result members as the reference, so that the body of the reference's
// driver (src/main.cpp:52-77: `X bvh; bvh.build(context, triangles); bvh.traverseBvh(context);`) compiles and runs against this
// header (examples/reference_driver.cpp is that body; tests/test_cpp_mirror.py runs it on the GPU and diffs its trees with the ctypes path):
//   BvhConstruction::{TwoPassLbvh, SinglePassLbvh, PLOCNew, HPLOC}::build(Context&, std::vector<Triangle>&) and ::traverseBvh(Context&)
//   (src/TwoPassLbvh.h:12-32, src/SinglePassLbvh.h:12-32, src/PLOC++Bvh.h:12-33, src/Hploc.h:12-33).
// build() does what the reference's build() does on the device: E, M, S, B, then CollapseToWide4Bvh, and m_cost = the BVH4 cost of
// Utility::calculatebvh4Cost (src/TwoPassLbvh.cpp:154-197).  traverseBvh() is the reference's per-builder flavour: TwoPassLbvh renders
// 512x512 with the speculative while-while kernel and transformation (0,0,-5)/1 (src/TwoPassLbvh.cpp:199-311), SinglePassLbvh with the
// if-if kernel and (0,0,-3)/3 (src/SinglePassLbvh.cpp:190-311: IFIF wins its #if chain), PLOCNew / HPLOC only print (src/PLOC++Bvh.cpp:198-212,
// src/Hploc.cpp:167-181); all print the reference's perf block to std::cout.
// Differences, all documented in INTEGRATION.md:
//   * d_* members are lightweight views (ptr()/size()/getData()) of ctx-owned device memory instead of Oro::GpuMemory; they stay valid
//     until the next build on the same Context.  d_mortonCodeValues (value i = i) is materialised on the first ptr() call after a build; its getData() returns the iota;
//   * Context owns a bvh_ctx (device + stream + arena) instead of an Orochi context; device selectable (reference: 0);
//   * SinglePassLbvh keeps m_rootNodeIdx = the BVH2 root (the reference overwrites it with the BVH4 root 0 before the BVH2 traversal
//     uses it, src/SinglePassLbvh.cpp:183 vs :265 — a reference bug); the BVH4 root is always 0;
//   * the image stays in m_colorBuffer (RGBA8, the reference's d_colorBuffer read-back); no PNG is written (stb is out of scope);
//   * errors: build() throws std::runtime_error with the C-ABI code (the reference prints and continues).
// Header-only; link with -lbvh_mi355x.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <numeric>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../bvh_mi355x.h"

namespace BvhConstruction {

using u32 = uint32_t; using u8 = uint8_t; using u64 = uint64_t;
constexpr u32 INVALID_NODE_IDX = BVH_INVALID;   // src/Common.h:90
constexpr u32 INVALID_PRIM_IDX = BVH_INVALID;   // src/Common.h:91
constexpr float FltMax = BVH_FLT_MAX;           // src/Common.h:86
constexpr float Pi = 3.14159265358979323846f;

struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct Aabb { float3 m_min{FltMax, FltMax, FltMax}, m_max{-FltMax, -FltMax, -FltMax};   // src/Common.h:310-416 (data + the helpers hosts use)
    float3 extent() const { return {m_max.x - m_min.x, m_max.y - m_min.y, m_max.z - m_min.z}; }
    float area() const { const float3 e = extent(); return 2 * (e.x * e.y + e.x * e.z + e.y * e.z); } };
struct alignas(64) Triangle { float3 v1, v2, v3; };                                        // src/Common.h:429-434
struct alignas(32) Bvh2Node { u32 m_leftChildIdx, m_rightChildIdx; Aabb m_aabb; };         // src/Common.h:436-441
struct PrimRef { u32 m_primIdx = INVALID_PRIM_IDX; Aabb m_aabb; };                         // src/Common.h:574-578
struct alignas(32) Ray { float3 m_origin; float3 m_direction; float m_tMin = 0.0f; float m_tMax = FltMax; };                     // :533-539
struct alignas(64) Transformation { float3 m_translation; float m_pad; float3 m_scale; float m_pad1; float4 m_quat; };           // :541-548
struct alignas(64) Camera { float4 m_eye; float4 m_quat; float m_fov; float m_near; float m_far; float m_pad; };                 // :550-558
struct alignas(128) Bvh4Node { Aabb m_aabb[4]; u32 m_child[4] = {INVALID_NODE_IDX, INVALID_NODE_IDX, INVALID_NODE_IDX, INVALID_NODE_IDX};
                               u32 m_parent = INVALID_NODE_IDX; u32 m_childCount = 2; };                                         // :560-566
struct PrimNode { u32 m_primIdx = INVALID_PRIM_IDX; u32 m_parent = INVALID_NODE_IDX; };                                          // :568-572
static_assert(sizeof(Aabb) == 24 && sizeof(Triangle) == 64 && sizeof(Bvh2Node) == 32 && sizeof(PrimRef) == 28, "reference ABI");
static_assert(sizeof(Ray) == 32 && sizeof(Transformation) == 64 && sizeof(Camera) == 64 && sizeof(Bvh4Node) == 128 && sizeof(PrimNode) == 8, "reference ABI");

inline float4 qtGetIdentity() { return float4{0.0f, 0.0f, 0.0f, 1.0f}; }                  // src/Common.h:474
inline float4 qtRotation(float4 axisAngle) {                                               // src/Common.h:461-472
    const float len = std::sqrt(axisAngle.x * axisAngle.x + axisAngle.y * axisAngle.y + axisAngle.z * axisAngle.z);
    const float s = std::sin(axisAngle.w / 2.0f), c = std::cos(axisAngle.w / 2.0f);
    return float4{axisAngle.x / len * s, axisAngle.y / len * s, axisAngle.z / len * s, c};
}

enum TimerCodes { CalculateCentroidExtentsTime, CalculateMortonCodesTime, SortingTime, BvhBuildTime, TraversalTime, CollapseBvhTime, RayGenTime };  // src/Common.h:418-427

inline void check(int rc, const char* what) { if (rc != 0) throw std::runtime_error(std::string(what) + " failed: " + std::to_string(rc)); }

class Context {                                   // src/Context.h:8-18
public:
    explicit Context(int device = 0) { check(bvh_ctx_create(device, &m_ctx), "bvh_ctx_create"); bvh_ctx_set_profiling(m_ctx, 1); }
    ~Context() { bvh_ctx_destroy(m_ctx); }
    Context(const Context&) = delete; Context& operator=(const Context&) = delete;
    bvh_ctx* handle() const { return m_ctx; }
private:
    bvh_ctx* m_ctx = nullptr;
};

class Timer {                                     // src/Timer.h:31-73 (read side)
public:
    float getTimeRecord(int token) const { auto it = rec.find(token); return it == rec.end() ? 0.f : it->second; }
    void set(int token, float ms) { rec[token] = ms; }
private:
    std::unordered_map<int, float> rec;
};

// view of a ctx-owned device array with the read-back call the reference's hosts use (Oro::GpuMemory<T>::getData)
template <typename T> class DeviceView {
public:
    void bind(bvh_ctx* c, const void* p, size_t n) { m_ctx = c; m_ptr = static_cast<T*>(const_cast<void*>(p)); m_size = n; }
    T* ptr() const { return m_ptr; }
    size_t size() const { return m_size; }
    std::vector<T> getData() const { std::vector<T> h(m_size); if (m_size) check(bvh_dev_download(m_ctx, h.data(), m_ptr, m_size * sizeof(T)), "getData"); return h; }
private:
    bvh_ctx* m_ctx = nullptr; T* m_ptr = nullptr; size_t m_size = 0;
};
// d_mortonCodeValues: value i = i (src/CommonBlocksKernel.h:384); this pipeline produces the indices inside the first sort pass instead of
// writing and re-reading 4 bytes per primitive, so the build leaves no device array behind it.  A host that keeps one of the reference's stages beside the
// mirror hands `d_mortonCodeValues.ptr()` to it (Oro::RadixSort::sort's source values, src/Hploc.cpp:63-81): ptr() therefore materialises the iota on its first
// call after a build — one allocation owned by the view (re-used by later builds of at most that size) and one upload — instead of returning null.
class IotaView {
public:
    IotaView() = default;
    IotaView(const IotaView&) = delete; IotaView& operator=(const IotaView&) = delete;
    ~IotaView() { if (m_ptr) bvh_dev_free(m_ctx, m_ptr); }
    void bind(bvh_ctx* c, size_t n) {
        if (m_ptr && (c != m_ctx || n > m_cap)) { bvh_dev_free(m_ctx, m_ptr); m_ptr = nullptr; m_cap = 0; }
        m_ctx = c; m_size = n; m_filled = 0;
    }
    u32* ptr() const {
        if (!m_size) return nullptr;
        if (!m_ptr) { void* p = nullptr; check(bvh_dev_alloc(m_ctx, m_size * sizeof(u32), &p), "bvh_dev_alloc"); m_ptr = static_cast<u32*>(p); m_cap = m_size; m_filled = 0; }
        if (m_filled < m_size) { const std::vector<u32> h = getData(); check(bvh_dev_upload(m_ctx, m_ptr, h.data(), m_size * sizeof(u32)), "bvh_dev_upload"); m_filled = m_size; }
        return m_ptr;
    }
    size_t size() const { return m_size; }
    std::vector<u32> getData() const { std::vector<u32> h(m_size); std::iota(h.begin(), h.end(), 0u); return h; }
private:
    bvh_ctx* m_ctx = nullptr; size_t m_size = 0;
    mutable u32* m_ptr = nullptr; mutable size_t m_cap = 0, m_filled = 0;      // (an iota of m_filled entries is a prefix of every longer one: a smaller rebuild keeps it)
};
// a device allocation owned by the builder object (the wide tree, the image buffers)
template <typename T> class DeviceArray {
public:
    DeviceArray() = default;
    DeviceArray(const DeviceArray&) = delete; DeviceArray& operator=(const DeviceArray&) = delete;
    ~DeviceArray() { reset(); }
    void resize(bvh_ctx* c, size_t n) { if (n <= m_cap && c == m_ctx) { m_size = n; return; } reset(); m_ctx = c; void* p = nullptr; check(bvh_dev_alloc(c, n * sizeof(T), &p), "bvh_dev_alloc"); m_ptr = static_cast<T*>(p); m_cap = m_size = n; }
    void reset() { if (m_ptr) bvh_dev_free(m_ctx, m_ptr); m_ptr = nullptr; m_cap = m_size = 0; }
    T* ptr() const { return m_ptr; }
    size_t size() const { return m_size; }
    std::vector<T> getData(size_t count) const { std::vector<T> h(count); if (count) check(bvh_dev_download(m_ctx, h.data(), m_ptr, count * sizeof(T)), "getData"); return h; }
    std::vector<T> getData() const { return getData(m_size); }
private:
    bvh_ctx* m_ctx = nullptr; T* m_ptr = nullptr; size_t m_cap = 0, m_size = 0;
};

namespace detail {
// what each reference builder's traverseBvh() does (file:line in the header comment)
struct TraverseFlavour { bool render; bvh_trace_kind kind; float3 translation; float scale; };
template <bvh_algo ALGO> constexpr TraverseFlavour flavour() {
    return ALGO == BVH_LBVH_TWOPASS    ? TraverseFlavour{true, BVH_TRACE_SPECULATIVE_WHILE, {0.0f, 0.0f, -5.0f}, 1.0f}
         : ALGO == BVH_LBVH_SINGLEPASS ? TraverseFlavour{true, BVH_TRACE_IF_IF, {0.0f, 0.0f, -3.0f}, 3.0f}
                                       : TraverseFlavour{false, BVH_TRACE_WHILE_WHILE, {0.0f, 0.0f, 0.0f}, 1.0f};
}

template <bvh_algo ALGO> class Builder {
public:
    void build(Context& context, std::vector<Triangle>& primitives) {
        const u32 n = static_cast<u32>(primitives.size());
        bvh_result r{}; bvh_timings t{};
        check(bvh_build(context.handle(), ALGO, primitives.data(), n, 0, &r, &t), "build");
        m_triFormat = BVH_TRI_PADDED64;
        publish(context, r, t, n);
    }
    // beyond the reference (bvh_build_ex): device-resident input in any bvh_tri_format, 30- or 60-bit Morton codes.  With 60-bit codes
    // d_sortedMortonCodeKeys64 is bound instead of d_sortedMortonCodeKeys.
    void build(Context& context, const bvh_build_input& input, u32 n) {
        bvh_result r{}; bvh_timings t{};
        check(bvh_build_ex(context.handle(), ALGO, &input, n, &r, &t), "build_ex");
        m_triFormat = input.tri_format;
        publish(context, r, t, n);
    }

    // X::traverseBvh(Context&): GenerateRays -> the traversal kernel this builder's reference source selects -> RGBA read-back -> perf block.
    void traverseBvh(Context& context) {
        constexpr TraverseFlavour f = flavour<ALGO>();
        if (f.render && m_result.d_nodes && m_triFormat == BVH_TRI_PADDED64) {
            Transformation t{}; t.m_translation = f.translation; t.m_scale = float3{f.scale, f.scale, f.scale}; t.m_quat = qtGetIdentity();
            Camera cam{}; cam.m_eye = float4{0.0f, 2.5f, 5.8f, 0.0f}; cam.m_quat = qtRotation(float4{0.0f, 0.0f, 1.0f, -1.57f});
            cam.m_fov = 45.0f * Pi / 180.f; cam.m_near = 0.0f; cam.m_far = 100000.0f;
            render(context, cam, t, f.kind, 512, 512);
        }
        printPerf(std::cout);
    }
    // the image part of traverseBvh with caller-chosen view and kernel (any builder; PLOC layouts go through the LBVH-layout adapter)
    void render(Context& context, const Camera& cam, const Transformation& t, bvh_trace_kind kind, u32 width, u32 height) {
        bvh_ctx* c = context.handle();
        const u32 n = m_result.n_leaves;
        d_rayBuffer.resize(c, size_t(width) * height); d_rayCounterBuffer.resize(c, size_t(width) * height); d_colorBuffer.resize(c, size_t(width) * height * 4);
        const void* nodes = m_result.d_nodes;
        if (m_result.layout == 1) { d_lbvhLayoutNodes.resize(c, 2 * size_t(n) - 1); check(bvh_to_lbvh_layout(c, &m_result, d_lbvhLayoutNodes.ptr()), "bvh_to_lbvh_layout"); nodes = d_lbvhLayoutNodes.ptr(); }
        using clk = std::chrono::steady_clock;           // (Timer::measure brackets each launch with a blocking wait, src/Timer.h:50-52)
        check(bvh_ctx_synchronize(c), "sync"); const auto t0 = clk::now();
        check(bvh_generate_rays(c, &cam, d_rayBuffer.ptr(), width, height), "GenerateRays");
        check(bvh_ctx_synchronize(c), "sync"); const auto t1 = clk::now();
        check(bvh_trace(c, kind, d_rayBuffer.ptr(), m_result.d_tris, nodes, m_rootNodeIdx, m_nInternalNodes, &t, d_colorBuffer.ptr(), d_rayCounterBuffer.ptr(), width, height), "traversal");
        check(bvh_ctx_synchronize(c), "sync"); const auto t2 = clk::now();
        m_timer.set(RayGenTime, std::chrono::duration<float, std::milli>(t1 - t0).count());
        m_timer.set(TraversalTime, std::chrono::duration<float, std::milli>(t2 - t1).count());
        m_colorBuffer = d_colorBuffer.getData();
        m_width = width; m_height = height;
    }
    void printPerf(std::ostream& os) const {      // src/TwoPassLbvh.cpp:300-310
        os << "==========================Perf Times==========================" << std::endl;
        os << "CalculateCentroidExtentsTime :" << m_timer.getTimeRecord(CalculateCentroidExtentsTime) << "ms" << std::endl;
        os << "CalculateMortonCodesTime :" << m_timer.getTimeRecord(CalculateMortonCodesTime) << "ms" << std::endl;
        os << "SortingTime : " << m_timer.getTimeRecord(SortingTime) << "ms" << std::endl;
        os << "BvhBuildTime : " << m_timer.getTimeRecord(BvhBuildTime) << "ms" << std::endl;
        os << "TraversalTime : " << m_timer.getTimeRecord(TraversalTime) << "ms" << std::endl;
        os << "CollapseTime : " << m_timer.getTimeRecord(CollapseBvhTime) << "ms" << std::endl;
        os << "Bvh Cost : " << m_cost << std::endl;
        os << "Total Time : " << m_timer.getTimeRecord(CalculateCentroidExtentsTime) + m_timer.getTimeRecord(CalculateMortonCodesTime) +
                  m_timer.getTimeRecord(SortingTime) + m_timer.getTimeRecord(BvhBuildTime) << "ms" << std::endl;
        os << "==============================================================" << std::endl;
    }
private:
    void publish(Context& context, const bvh_result& r, const bvh_timings& t, u32 n) {
        bvh_ctx* c = context.handle();
        m_result = r;
        const size_t nodes = r.layout == 0 ? 2 * size_t(n) - 1 : size_t(n) - 1;
        d_triangleBuff.bind(c, m_triFormat == BVH_TRI_PADDED64 ? r.d_tris : nullptr, m_triFormat == BVH_TRI_PADDED64 ? n : 0);
        d_bvhNodes.bind(c, r.d_nodes, nodes);
        d_leafNodes.bind(c, r.d_leaves, r.d_leaves ? n : 0);
        d_triangleAabb.bind(c, r.d_prim_aabbs, n);
        d_sceneExtents.bind(c, r.d_scene_extent, 1);
        d_sortedMortonCodeKeys.bind(c, r.key_bits == 64 ? nullptr : r.d_sorted_keys, r.key_bits == 64 ? 0 : n);
        d_sortedMortonCodeKeys64.bind(c, r.key_bits == 64 ? r.d_sorted_keys : nullptr, r.key_bits == 64 ? n : 0);
        d_sortedMortonCodeValues.bind(c, r.d_sorted_vals, n);
        d_mortonCodeKeys.bind(c, r.key_bits == 64 ? nullptr : r.d_morton_keys, r.key_bits == 64 ? 0 : n);
        d_mortonCodeKeys64.bind(c, r.key_bits == 64 ? r.d_morton_keys : nullptr, r.key_bits == 64 ? n : 0);
        d_mortonCodeValues.bind(c, n);
        m_rootNodeIdx = r.root; m_nInternalNodes = r.n_internal;
        m_timer.set(CalculateCentroidExtentsTime, t.ms_extents); m_timer.set(CalculateMortonCodesTime, t.ms_morton);
        m_timer.set(SortingTime, t.ms_sort); m_timer.set(BvhBuildTime, t.ms_build);
        double c2 = 0; check(bvh_sah_cost(c, &r, &c2), "bvh_sah_cost"); m_costBvh2 = static_cast<float>(c2);
        // CollapseToWide4Bvh + m_cost = calculatebvh4Cost (src/TwoPassLbvh.cpp:154-197)
        d_wideBvhNodes.resize(c, n); d_wideLeafNodes.resize(c, n);
        check(bvh_collapse4(c, &r, d_wideBvhNodes.ptr(), d_wideLeafNodes.ptr(), &m_nWideNodes), "bvh_collapse4");
        float cms = 0.f; bvh_ctx_last_collapse_ms(c, &cms); m_timer.set(CollapseBvhTime, cms);
        double c4 = 0; check(bvh_bvh4_cost(c, d_wideBvhNodes.ptr(), m_nWideNodes, d_wideLeafNodes.ptr(), r.d_prim_aabbs, n, &c4), "bvh_bvh4_cost");
        m_cost = static_cast<float>(c4);
    }
    u32 m_triFormat = BVH_TRI_PADDED64;
public:
    // the reference's public members (src/Hploc.h:19-32, src/TwoPassLbvh.h:19-31)
    DeviceView<Triangle> d_triangleBuff;                                   // (bound for 64-byte Triangle inputs only)
    DeviceView<Aabb> d_triangleAabb, d_sceneExtents;
    DeviceView<u32> d_mortonCodeKeys;
    IotaView d_mortonCodeValues;
    DeviceView<u32> d_sortedMortonCodeKeys, d_sortedMortonCodeValues;
    DeviceView<uint64_t> d_mortonCodeKeys64, d_sortedMortonCodeKeys64;     // 60-bit builds (beyond the reference)
    DeviceView<Bvh2Node> d_bvhNodes;
    DeviceView<PrimRef> d_leafNodes;
    DeviceView<u32> d_flags;             // src/TwoPassLbvh.h:27, src/SinglePassLbvh.h:27: the reference's refit counters (scratch of its BvhBuild / FitBvhNodes launches).  This
                                         // pipeline's parent-claim words are self-cleaning context scratch with another meaning, so the view is EMPTY (size() == 0): a host that
                                         // names the member compiles, one that reads counters out of it gets none
    u32 m_rootNodeIdx = 0;
    Timer m_timer;
    u32 m_nInternalNodes = 0;
    float m_cost = 0.0f;                 // BVH4 cost (Utility::calculatebvh4Cost), as in the reference after build()
    // beyond the reference's members
    float m_costBvh2 = 0.0f;             // BVH2 SAH (Utility::calculateLbvhCost formula)
    u32 m_nWideNodes = 0;                // wide nodes of the collapsed tree (root 0)
    DeviceArray<Bvh4Node> d_wideBvhNodes; DeviceArray<PrimNode> d_wideLeafNodes;      // locals of the reference's build(), kept here
    DeviceArray<Ray> d_rayBuffer; DeviceArray<u32> d_rayCounterBuffer; DeviceArray<u8> d_colorBuffer; DeviceArray<Bvh2Node> d_lbvhLayoutNodes;
    std::vector<u8> m_colorBuffer; u32 m_width = 0, m_height = 0;                      // the image traverseBvh() rendered (RGBA8)
    bvh_result m_result{};
};
}  // namespace detail

// src/BatchedBuilder.h:12-31, re-purposed as the scene shard of BASELINE.json config 5: one mesh per GPU, RCCL all-gather of roots.
// The per-device contexts and the communicator live as long as the object (bvh_batch).
// every mesh's array of one kind (nodes or leaves), each on the device that built the mesh: what the reference keeps in ONE Oro::GpuMemory (src/BatchedBuilder.h:24-25)
template <typename T> class BatchView {
public:
    struct Segment { int device; const T* ptr; size_t count; };
    void clear() { m_seg.clear(); m_batch = nullptr; }
    void bind(bvh_batch* b, const std::vector<bvh_batch_mesh>* meshes, bool leaves) {
        m_batch = b; m_meshes = meshes; m_leaves = leaves; m_seg.clear();
        for (const auto& m : *meshes) m_seg.push_back(Segment{m.device, static_cast<const T*>(leaves ? m.d_leaves : m.d_nodes), leaves ? (m.d_leaves ? size_t(m.n_leaves) : 0) : size_t(m.n_nodes)});
    }
    size_t size() const { size_t t = 0; for (const auto& s : m_seg) t += s.count; return t; }      // all meshes together
    size_t meshes() const { return m_seg.size(); }
    const Segment& segment(size_t m) const { return m_seg[m]; }                                    // mesh m's part (device pointer valid on segment(m).device)
    T* ptr() const { return m_seg.empty() ? nullptr : const_cast<T*>(m_seg[0].ptr); }              // mesh 0's part
    std::vector<T> getData(size_t m) const {                                                       // one mesh
        std::vector<T> h(m_seg[m].count);
        if (!h.empty()) check(bvh_batch_download(m_batch, &(*m_meshes)[m], m_leaves ? nullptr : h.data(), m_leaves ? h.data() : nullptr), "bvh_batch_download");
        return h;
    }
    std::vector<T> getData() const {                                                               // concatenated in mesh order, like the reference's single array
        std::vector<T> all; all.reserve(size());
        for (size_t m = 0; m < m_seg.size(); ++m) { const auto part = getData(m); all.insert(all.end(), part.begin(), part.end()); }
        return all;
    }
private:
    bvh_batch* m_batch = nullptr; const std::vector<bvh_batch_mesh>* m_meshes = nullptr; bool m_leaves = false; std::vector<Segment> m_seg;
};
// d_rootNodes: one root index per mesh (host-resident here: the indices are known to the host the moment a build returns)
class RootView {
public:
    void assign(std::vector<u32> v) { m_v = std::move(v); }
    u32* ptr() const { return nullptr; }
    size_t size() const { return m_v.size(); }
    std::vector<u32> getData() const { return m_v; }
private:
    std::vector<u32> m_v;
};
struct BatchedBuildInput { std::vector<Triangle> m_primitives; };
class BatchedBvhBuilder {
public:
    explicit BatchedBvhBuilder(std::vector<int> devices = {0}, bvh_algo algo = BVH_HPLOC) : m_devices(std::move(devices)), m_algo(algo) {}
    ~BatchedBvhBuilder() { if (m_batch) bvh_batch_destroy(m_batch); }
    BatchedBvhBuilder(const BatchedBvhBuilder&) = delete; BatchedBvhBuilder& operator=(const BatchedBvhBuilder&) = delete;
    void build(Context&, std::vector<BatchedBuildInput>& batch) {
        if (!m_batch) check(bvh_batch_create((int)m_devices.size(), m_devices.data(), &m_batch), "bvh_batch_create");
        std::vector<const void*> ptrs; std::vector<uint32_t> counts;
        for (auto& b : batch) { ptrs.push_back(b.m_primitives.data()); counts.push_back((uint32_t)b.m_primitives.size()); }
        m_rootAabbs.assign(batch.size(), Aabb{}); m_buildMs.assign(batch.size(), 0.f); m_checksums.assign(batch.size(), 0);
        m_sah.assign(batch.size(), 0.0); m_meshes.assign(batch.size(), bvh_batch_mesh{});
        bvh_batch_report rep{}; rep.root_aabbs = reinterpret_cast<float*>(m_rootAabbs.data()); rep.build_ms = m_buildMs.data(); rep.checksums = m_checksums.data();
        rep.sah = m_sah.data(); rep.meshes = m_meshes.data();
        check(bvh_batch_build(m_batch, m_algo, ptrs.data(), counts.data(), (int)batch.size(), &rep), "bvh_batch_build");
        m_allGatherUs = rep.allgather_us; m_wallMs = rep.wall_ms; m_lanesPerDevice = rep.lanes_per_device;
        // the reference's members (src/BatchedBuilder.h:24-30)
        d_bvhNodes.bind(m_batch, &m_meshes, false); d_primRefs.bind(m_batch, &m_meshes, true);
        std::vector<u32> roots; size_t off = 0; m_nInternalNodes = 0; double cost = 0; float dev_ms = 0.f;
        std::vector<float> per_dev(m_devices.size(), 0.f);
        for (size_t m = 0; m < m_meshes.size(); ++m) {
            roots.push_back(u32(off + m_meshes[m].root)); off += m_meshes[m].n_nodes;               // index into the concatenated d_bvhNodes.getData()
            m_nInternalNodes += m_meshes[m].n_internal; cost += m_sah[m];
            per_dev[m % m_devices.size()] += m_buildMs[m];
        }
        for (float v : per_dev) dev_ms = v > dev_ms ? v : dev_ms;
        d_rootNodes.assign(roots);
        m_rootNodeIdx = roots.empty() ? 0u : roots[0];
        m_cost = m_meshes.empty() ? 0.f : float(cost / double(m_meshes.size()));
        m_timer.set(BvhBuildTime, dev_ms);
    }
    void traverseBvh(Context&) {   // (the reference's flavour renders one of its tiny trees; here: the per-mesh report)
        for (size_t m = 0; m < m_buildMs.size(); ++m) std::cout << "mesh " << m << " build " << m_buildMs[m] << "ms" << std::endl;
        std::cout << "root AABB all-gather " << m_allGatherUs << "us" << std::endl;
    }
    // the reference's public members (src/BatchedBuilder.h:24-30).  Child indices inside a mesh's nodes are mesh-local; d_rootNodes[m] is mesh m's root as an index
    // into the concatenated getData() of d_bvhNodes (= the mesh's node offset + its local root)
    BatchView<Bvh2Node> d_bvhNodes;     // every mesh's Bvh2Node array (PLOC layouts: n-1 internal nodes; LBVH: 2n-1 with the leaves inside)
    BatchView<PrimRef> d_primRefs;      // every mesh's PrimRef leaves (PLOC layouts; empty for the LBVH builders)
    RootView d_rootNodes;
    u32 m_rootNodeIdx = 0;              // = d_rootNodes[0]
    Timer m_timer;                      // BvhBuildTime: device time of the batch = max over devices of the sum of its meshes' E+M+S+B
    u32 m_nInternalNodes = 0;           // of all meshes together
    float m_cost = 0.0f;                // mean BVH2 SAH cost of the batch's meshes (the reference never sets it)
    // beyond the reference's members
    std::vector<Aabb> m_rootAabbs;      // TLAS input: one root box per mesh, identical on every device after the all-gather
    std::vector<float> m_buildMs;
    std::vector<uint64_t> m_checksums;  // bvh_checksum of every mesh's tree
    std::vector<double> m_sah;
    std::vector<bvh_batch_mesh> m_meshes;
    float m_allGatherUs = 0.f, m_wallMs = 0.f; int m_lanesPerDevice = 0;
private:
    std::vector<int> m_devices; bvh_algo m_algo; bvh_batch* m_batch = nullptr;
};

class TwoPassLbvh : public detail::Builder<BVH_LBVH_TWOPASS> {};        // src/TwoPassLbvh.h:12-32
class SinglePassLbvh : public detail::Builder<BVH_LBVH_SINGLEPASS> {};  // src/SinglePassLbvh.h:12-32
class PLOCNew : public detail::Builder<BVH_PLOCPP> {};                  // src/PLOC++Bvh.h:12-33
class HPLOC : public detail::Builder<BVH_HPLOC> {};                     // src/Hploc.h:12-33

}  // namespace BvhConstruction
