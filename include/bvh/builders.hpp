// builders.hpp — C++ host mirror of the reference's builder API over the C ABI (include/bvh_mi355x.h).
//
// Same namespace, class names, method signatures and public result members as the reference, so that a caller of
//   BvhConstruction::{TwoPassLbvh, SinglePassLbvh, PLOCNew, HPLOC}::build(Context&, std::vector<Triangle>&)
// (reference src/TwoPassLbvh.h:12-32, src/SinglePassLbvh.h:12-32, src/PLOC++Bvh.h:12-33, src/Hploc.h:12-33; driver
// src/main.cpp:52-65) recompiles against this header unchanged.  Differences, all documented in INTEGRATION.md:
//   * d_* members are lightweight views (ptr()/size()/getData()) of ctx-owned device memory instead of Oro::GpuMemory;
//   * Context owns a bvh_ctx (device + stream + arena) instead of an Orochi context; device selectable (reference: 0);
//   * SinglePassLbvh exposes m_rootNodeIdx = the BVH2 root (the reference overwrites it with the BVH4 root 0 before the
//     BVH2 traversal uses it, src/SinglePassLbvh.cpp:183 vs :265 — a reference bug);
//   * errors: build() throws std::runtime_error with the C-ABI code (the reference prints and continues).
// Header-only; link with -lbvh_mi355x.
#pragma once
#include <cstdint>
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

struct float3 { float x, y, z; };
struct Aabb { float3 m_min{FltMax, FltMax, FltMax}, m_max{-FltMax, -FltMax, -FltMax};   // src/Common.h:310-416 (data + the helpers hosts use)
    float3 extent() const { return {m_max.x - m_min.x, m_max.y - m_min.y, m_max.z - m_min.z}; }
    float area() const { const float3 e = extent(); return 2 * (e.x * e.y + e.x * e.z + e.y * e.z); } };
struct alignas(64) Triangle { float3 v1, v2, v3; };                                        // src/Common.h:429-434
struct alignas(32) Bvh2Node { u32 m_leftChildIdx, m_rightChildIdx; Aabb m_aabb; };         // src/Common.h:436-441
struct PrimRef { u32 m_primIdx = INVALID_PRIM_IDX; Aabb m_aabb; };                         // src/Common.h:574-578
static_assert(sizeof(Aabb) == 24 && sizeof(Triangle) == 64 && sizeof(Bvh2Node) == 32 && sizeof(PrimRef) == 28, "reference ABI");

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
    void bind(bvh_ctx* c, void* p, size_t n) { m_ctx = c; m_ptr = static_cast<T*>(p); m_size = n; }
    T* ptr() const { return m_ptr; }
    size_t size() const { return m_size; }
    std::vector<T> getData() const { std::vector<T> h(m_size); if (m_size) check(bvh_dev_download(m_ctx, h.data(), m_ptr, m_size * sizeof(T)), "getData"); return h; }
private:
    bvh_ctx* m_ctx = nullptr; T* m_ptr = nullptr; size_t m_size = 0;
};

namespace detail {
template <bvh_algo ALGO> class Builder {
public:
    void build(Context& context, std::vector<Triangle>& primitives) {
        const u32 n = static_cast<u32>(primitives.size());
        bvh_result r{}; bvh_timings t{};
        check(bvh_build(context.handle(), ALGO, primitives.data(), n, 0, &r, &t), "build");
        publish(context, r, t, n);
    }
    // beyond the reference (bvh_build_ex): device-resident input in any bvh_tri_format, 30- or 60-bit Morton codes.  With 60-bit codes
    // d_sortedMortonCodeKeys64 is bound instead of d_sortedMortonCodeKeys.
    void build(Context& context, const bvh_build_input& input, u32 n) {
        bvh_result r{}; bvh_timings t{};
        check(bvh_build_ex(context.handle(), ALGO, &input, n, &r, &t), "build_ex");
        publish(context, r, t, n);
    }
private:
    void publish(Context& context, const bvh_result& r, const bvh_timings& t, u32 n) {
        m_result = r;
        const size_t nodes = r.layout == 0 ? 2 * size_t(n) - 1 : size_t(n) - 1;
        d_bvhNodes.bind(context.handle(), r.d_nodes, nodes);
        d_leafNodes.bind(context.handle(), r.d_leaves, r.d_leaves ? n : 0);
        d_triangleAabb.bind(context.handle(), r.d_prim_aabbs, n);
        d_sceneExtents.bind(context.handle(), r.d_scene_extent, 1);
        d_sortedMortonCodeKeys.bind(context.handle(), r.key_bits == 64 ? nullptr : r.d_sorted_keys, r.key_bits == 64 ? 0 : n);
        d_sortedMortonCodeKeys64.bind(context.handle(), r.key_bits == 64 ? r.d_sorted_keys : nullptr, r.key_bits == 64 ? n : 0);
        d_sortedMortonCodeValues.bind(context.handle(), r.d_sorted_vals, n);
        m_rootNodeIdx = r.root; m_nInternalNodes = r.n_internal;
        m_timer.set(CalculateCentroidExtentsTime, t.ms_extents); m_timer.set(CalculateMortonCodesTime, t.ms_morton);
        m_timer.set(SortingTime, t.ms_sort); m_timer.set(BvhBuildTime, t.ms_build); m_timer.set(CollapseBvhTime, t.ms_collapse);
        double c = 0; check(bvh_sah_cost(context.handle(), &r, &c), "bvh_sah_cost"); m_cost = static_cast<float>(c);   // BVH2 SAH (the reference reports the BVH4 cost)
    }
public:
    // the reference's traverseBvh() prints the perf block (src/TwoPassLbvh.cpp:300-310); the PLOC/HPLOC flavours do nothing else
    std::string perfReport() const {
        auto f = [&](int tok) { return std::to_string(m_timer.getTimeRecord(tok)); };
        const float total = m_timer.getTimeRecord(CalculateCentroidExtentsTime) + m_timer.getTimeRecord(CalculateMortonCodesTime) + m_timer.getTimeRecord(SortingTime) + m_timer.getTimeRecord(BvhBuildTime);
        return "CalculateCentroidExtentsTime :" + f(CalculateCentroidExtentsTime) + "ms\nCalculateMortonCodesTime :" + f(CalculateMortonCodesTime) + "ms\nSortingTime : " + f(SortingTime) +
               "ms\nBvhBuildTime : " + f(BvhBuildTime) + "ms\nBvh Cost : " + std::to_string(m_cost) + "\nTotal Time : " + std::to_string(total) + "ms\n";
    }
    DeviceView<Aabb> d_triangleAabb, d_sceneExtents;
    DeviceView<u32> d_sortedMortonCodeKeys, d_sortedMortonCodeValues;
    DeviceView<uint64_t> d_sortedMortonCodeKeys64;
    DeviceView<Bvh2Node> d_bvhNodes;
    DeviceView<PrimRef> d_leafNodes;
    u32 m_rootNodeIdx = 0;
    Timer m_timer;
    u32 m_nInternalNodes = 0;
    float m_cost = 0.0f;
    bvh_result m_result{};
};
}  // namespace detail

// src/BatchedBuilder.h:12-31, re-purposed as the scene shard of BASELINE.json config 5: one mesh per GPU, RCCL all-gather of roots
struct BatchedBuildInput { std::vector<Triangle> m_primitives; };
class BatchedBvhBuilder {
public:
    explicit BatchedBvhBuilder(std::vector<int> devices = {0}, bvh_algo algo = BVH_HPLOC) : m_devices(std::move(devices)), m_algo(algo) {}
    void build(Context&, std::vector<BatchedBuildInput>& batch) {
        std::vector<const void*> ptrs; std::vector<uint32_t> counts;
        for (auto& b : batch) { ptrs.push_back(b.m_primitives.data()); counts.push_back((uint32_t)b.m_primitives.size()); }
        m_rootAabbs.assign(batch.size(), Aabb{}); m_buildMs.assign(batch.size(), 0.f);
        check(bvh_batched_build((int)m_devices.size(), m_devices.data(), m_algo, ptrs.data(), counts.data(), (int)batch.size(),
                                reinterpret_cast<float*>(m_rootAabbs.data()), m_buildMs.data()), "bvh_batched_build");
    }
    std::vector<Aabb> m_rootAabbs;      // TLAS input: one root box per mesh, identical on every device after the all-gather
    std::vector<float> m_buildMs;
private:
    std::vector<int> m_devices; bvh_algo m_algo;
};

class TwoPassLbvh : public detail::Builder<BVH_LBVH_TWOPASS> {};        // src/TwoPassLbvh.h:12-32
class SinglePassLbvh : public detail::Builder<BVH_LBVH_SINGLEPASS> {};  // src/SinglePassLbvh.h:12-32
class PLOCNew : public detail::Builder<BVH_PLOCPP> {};                  // src/PLOC++Bvh.h:12-33
class HPLOC : public detail::Builder<BVH_HPLOC> {};                     // src/Hploc.h:12-33

}  // namespace BvhConstruction
