// batched.hip — scene-level multi-GPU shard inside ONE process (BASELINE.json config 5 through the C ABI).
//
// API name follows the reference's BatchedBvhBuilder::build(Context&, std::vector<BatchedBuildInput>&)
// (src/BatchedBuilder.h:12-31; its kernel builds many <=32-primitive trees on one GPU and does not compile — SURVEY.md
// Appendix B — so only the name, the "many independent meshes in, one result per mesh out" shape and the public result members
// — d_bvhNodes / d_primRefs / d_rootNodes: every mesh's nodes and leaves kept on the device, a root index per mesh — are kept).
// mesh m is built on devs[m % n_dev] by the ordinary single-GPU path (no peer traffic, no tree is ever split); afterwards ONE
// ncclAllGather of the per-device root-AABB slots (RCCL, xGMI) leaves the TLAS input on every device, and device devs[0]'s copy
// is returned to the host.  24 bytes per mesh: latency-only collective.
//
// Round 4 — a device that holds several meshes pipelines them: up to BATCH_LANES contexts per device (each
// its own stream, arena and triangle staging buffer), one host thread per context; a device's meshes are dealt to its lanes
// round-robin.  The H2D copy of mesh m+1 (pageable caller memory: the copy blocks its own host thread only), the build of mesh m
// and the checksum / SAH reduction / tree copy of mesh m-1 run on different streams and overlap on the device; round 3 built a
// device's meshes strictly one after another on one stream (13.8 ms of H2D + 1.3 ms of build per 10 M-triangle mesh, nothing
// overlapped).  With one mesh per device (config 5 on eight GPUs) nothing changes: one lane, one thread.
// bvh_batch keeps the contexts (arenas), the communicator, the gather buffers and the per-mesh output arenas across builds;
// bvh_batched_build is the one-shot form.
// (The multi-process flavour — one rank per GPU over torch.distributed — lives in the Python harness: batched.py, bench.py.)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <cstdio>
#include <chrono>
#include <new>
#include <cstring>
#include <thread>
#include <vector>
#include "bvh_mi355x.h"

namespace { constexpr int BATCH_LANES = 3; }

struct bvh_batch {
    std::vector<int> devs;
    std::vector<bvh_ctx*> ctx;           // [n_dev * BATCH_LANES]; lane 0 of every device exists from the start (its stream carries the collective),
                                         // the others are created when a device first holds that many meshes
    std::vector<ncclComm_t> comms;
    bool comm_ok = false;
    std::vector<float*> d_send, d_recv;
    std::vector<hipEvent_t> ev0, ev1;
    int slots = 0;                       // root-box slots per device the gather buffers are sized for
    std::vector<char*> d_out;            // per device: the trees of its meshes of the last build (only when the caller asked for them)
    std::vector<size_t> out_bytes;
    bvh_ctx*& lane(int d, int k) { return ctx[(size_t)d * BATCH_LANES + k]; }
};

namespace {
int batch_reserve_slots(bvh_batch* b, int slots) {
    if (slots <= b->slots) return 0;
    const int n_dev = (int)b->devs.size();
    b->slots = 0;                        // a failure below leaves some buffers freed: the next call must reallocate, whatever it asks for
    for (int d = 0; d < n_dev; ++d) {
        if (hipSetDevice(b->devs[d]) != hipSuccess) return BVH_E_INTERNAL;
        if (b->d_send[d]) (void)hipFree(b->d_send[d]);
        if (b->d_recv[d]) (void)hipFree(b->d_recv[d]);
        b->d_send[d] = b->d_recv[d] = nullptr;
        if (hipMalloc(&b->d_send[d], (size_t)slots * 6 * sizeof(float)) != hipSuccess) return BVH_E_INTERNAL;
        if (hipMalloc(&b->d_recv[d], (size_t)slots * n_dev * 6 * sizeof(float)) != hipSuccess) return BVH_E_INTERNAL;
    }
    b->slots = slots;
    return 0;
}
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }
// bytes of one mesh's tree in the output arena: nodes (layout 0: 2n-1 of them, leaves inside; layout 1: n-1) + PrimRef leaves (layout 1)
inline size_t tree_nodes(bvh_algo a, uint32_t n) { return (a == BVH_LBVH_TWOPASS || a == BVH_LBVH_SINGLEPASS) ? 2 * (size_t)n - 1 : (size_t)n - 1; }
inline size_t tree_leaves(bvh_algo a, uint32_t n) { return (a == BVH_LBVH_TWOPASS || a == BVH_LBVH_SINGLEPASS) ? 0 : (size_t)n; }
}  // namespace

extern "C" int bvh_batch_create(int n_dev, const int* devs, bvh_batch** out) {
    if (n_dev <= 0 || !devs || !out) return BVH_E_INVALID_ARG;
    *out = nullptr;
    bvh_batch* b = new (std::nothrow) bvh_batch();
    if (!b) return BVH_E_INTERNAL;
    b->devs.assign(devs, devs + n_dev);
    b->ctx.assign((size_t)n_dev * BATCH_LANES, nullptr); b->comms.resize(n_dev); b->d_send.assign(n_dev, nullptr); b->d_recv.assign(n_dev, nullptr);
    b->ev0.assign(n_dev, nullptr); b->ev1.assign(n_dev, nullptr); b->d_out.assign(n_dev, nullptr); b->out_bytes.assign(n_dev, 0);
    int rc = 0;
    for (int d = 0; d < n_dev && !rc; ++d) {
        rc = bvh_ctx_create(devs[d], &b->lane(d, 0));
        if (rc == 0) rc = bvh_ctx_set_option(b->lane(d, 0), BVH_OPT_PLOC_SCHEDULER, 1);     // lanes share the device: no reliance on co-residency (chunk tickets always)
        if (!rc && (hipSetDevice(devs[d]) != hipSuccess || hipEventCreate(&b->ev0[d]) != hipSuccess || hipEventCreate(&b->ev1[d]) != hipSuccess)) rc = BVH_E_INTERNAL;
    }
    if (!rc) { if (ncclCommInitAll(b->comms.data(), n_dev, devs) != ncclSuccess) rc = BVH_E_INTERNAL; else b->comm_ok = true; }
    if (rc) { bvh_batch_destroy(b); return rc; }
    *out = b;
    return 0;
}

extern "C" void bvh_batch_destroy(bvh_batch* b) {
    if (!b) return;
    const int n_dev = (int)b->devs.size();
    for (int d = 0; d < n_dev; ++d) {
        if (b->comm_ok) ncclCommDestroy(b->comms[d]);
        (void)hipSetDevice(b->devs[d]);
        if (b->d_send[d]) (void)hipFree(b->d_send[d]);
        if (b->d_recv[d]) (void)hipFree(b->d_recv[d]);
        if (b->d_out[d]) (void)hipFree(b->d_out[d]);
        if (b->ev0[d]) (void)hipEventDestroy(b->ev0[d]);
        if (b->ev1[d]) (void)hipEventDestroy(b->ev1[d]);
        for (int k = 0; k < BATCH_LANES; ++k) if (b->lane(d, k)) bvh_ctx_destroy(b->lane(d, k));
    }
    delete b;
}

extern "C" int bvh_batch_build(bvh_batch* b, bvh_algo algo, const void* const* h_tris, const uint32_t* n_tris, int n_meshes, bvh_batch_report* rep) {
    if (!b || !h_tris || !n_tris || n_meshes <= 0 || !rep || !rep->root_aabbs || (int)algo < 0 || (int)algo > 3) return BVH_E_INVALID_ARG;
    for (int m = 0; m < n_meshes; ++m) if (!h_tris[m] || n_tris[m] < 2) return BVH_E_INVALID_ARG;
    const auto t_begin = std::chrono::steady_clock::now();
    const int n_dev = (int)b->devs.size();
    const int slots = (n_meshes + n_dev - 1) / n_dev;
    int rc = batch_reserve_slots(b, slots); if (rc) return rc;
    const int lanes = slots < BATCH_LANES ? slots : BATCH_LANES;               // contexts (streams, host threads) per device in this call
    for (int d = 0; d < n_dev; ++d)
        for (int k = 1; k < lanes; ++k) if (!b->lane(d, k)) { rc = bvh_ctx_create(b->devs[d], &b->lane(d, k)); if (rc == 0) rc = bvh_ctx_set_option(b->lane(d, k), BVH_OPT_PLOC_SCHEDULER, 1); if (rc) return rc; }
    // per-mesh output slots (only when the caller wants the trees kept): device d's meshes one after another in its output arena
    std::vector<size_t> off_nodes(n_meshes, 0), off_leaves(n_meshes, 0);
    if (rep->meshes) {
        for (int d = 0; d < n_dev; ++d) {
            size_t total = 0;
            for (int m = d; m < n_meshes; m += n_dev) {
                off_nodes[m] = total; total += align256(tree_nodes(algo, n_tris[m]) * 32);
                off_leaves[m] = total; total += align256(tree_leaves(algo, n_tris[m]) * 28);
            }
            if (total > b->out_bytes[d]) {
                if (hipSetDevice(b->devs[d]) != hipSuccess) return BVH_E_INTERNAL;
                if (b->d_out[d]) { (void)hipFree(b->d_out[d]); b->d_out[d] = nullptr; b->out_bytes[d] = 0; }
                if (hipMalloc(&b->d_out[d], total) != hipSuccess) return BVH_E_INTERNAL;
                b->out_bytes[d] = total;
            }
        }
    }
    // slots without a mesh (n_meshes is not a multiple of the device count) gather zeros, not a previous call's boxes (blocking, before any lane starts: ADVICE r04)
    for (int d = 0; d < n_dev; ++d) if (hipSetDevice(b->devs[d]) != hipSuccess || hipMemset(b->d_send[d], 0, (size_t)slots * 6 * sizeof(float)) != hipSuccess) return BVH_E_INTERNAL;
    std::atomic<int> err{0};
    auto fail = [&](int code) { int z = 0; err.compare_exchange_strong(z, code); };
    // ---- builds: one host thread per (device, lane); device d's k-th mesh (m = d + k * n_dev) runs on lane k % lanes
    std::vector<std::thread> th;
    for (int d = 0; d < n_dev; ++d) for (int l = 0; l < lanes; ++l) th.emplace_back([&, d, l]() {
        bvh_ctx* c = b->lane(d, l);
        if (hipSetDevice(b->devs[d]) != hipSuccess) return fail(BVH_E_INTERNAL);
        // whatever way this lane leaves (its own error, another lane's), nothing of it is still in flight when bvh_batch_build returns: the next call may free and
        // re-allocate the output arena its copies write into
        struct Drain { bvh_ctx* c; ~Drain() { (void)bvh_ctx_synchronize(c); } } drain{c};
        bvh_ctx_set_profiling(c, rep->build_ms ? 1 : 0);
        for (int k = l; d + k * n_dev < n_meshes; k += lanes) {
            if (err.load()) return;
            const int m = d + k * n_dev;
            bvh_result r; bvh_timings t;
            int rc2 = bvh_build(c, algo, h_tris[m], n_tris[m], 0, &r, &t); if (rc2) return fail(rc2);
            if (rep->build_ms) rep->build_ms[m] = t.ms_total;
            if (r.root >= tree_nodes(algo, n_tris[m]) || !r.d_nodes) {
                return fail(BVH_E_INTERNAL);      // (never trust an index that is about to become an address)
            }
            // root AABB = nodes[root].aabb (24 bytes at offset 8 of the 32-byte node)
            rc2 = bvh_dev_copy(c, b->d_send[d] + 6 * k, (const char*)r.d_nodes + 32 * (size_t)r.root + 8, 24); if (rc2) return fail(rc2);
            if (rep->checksums) { rc2 = bvh_checksum(c, &r, &rep->checksums[m]); if (rc2) return fail(rc2); }
            if (rep->sah) { rc2 = bvh_sah_cost(c, &r, &rep->sah[m]); if (rc2) return fail(rc2); }
            if (rep->meshes) {           // the tree leaves the context's arena (the lane's next build reuses it)
                bvh_batch_mesh& o = rep->meshes[m];
                o.device = b->devs[d]; o.n_leaves = r.n_leaves; o.n_internal = r.n_internal; o.root = r.root; o.layout = r.layout;
                o.n_nodes = (uint32_t)tree_nodes(algo, n_tris[m]);
                o.d_nodes = b->d_out[d] + off_nodes[m]; o.d_leaves = r.layout == 1 ? b->d_out[d] + off_leaves[m] : nullptr;
                rc2 = bvh_dev_copy(c, (void*)o.d_nodes, r.d_nodes, (uint64_t)o.n_nodes * 32); if (rc2) return fail(rc2);
                if (o.d_leaves) { rc2 = bvh_dev_copy(c, (void*)o.d_leaves, r.d_leaves, (uint64_t)r.n_leaves * 28); if (rc2) return fail(rc2); }
            }
        }
        const int rc2 = bvh_ctx_synchronize(c); if (rc2) return fail(rc2);
    });
    for (auto& t : th) t.join();
    rc = err.load();
    if (rc && rep->meshes) std::memset(rep->meshes, 0, sizeof(bvh_batch_mesh) * (size_t)n_meshes);      // (no pointer into a half-written tree survives a failed call)
    // ---- the only collective: all-gather of the root-AABB slots (every lane's stream has been synchronised: the slots are written)
    if (!rc) {
        // RCCL enqueues the collective's kernels at ncclGroupEnd, not at the ncclAllGather call inside the group: the events that bracket it are
        // recorded on each device's stream before the group starts and after it ended (inside the group both would precede the collective)
        for (int d = 0; d < n_dev; ++d) { (void)hipSetDevice(b->devs[d]); (void)hipEventRecord(b->ev0[d], (hipStream_t)bvh_ctx_stream(b->lane(d, 0))); }
        ncclGroupStart();
        for (int d = 0; d < n_dev; ++d) {
            (void)hipSetDevice(b->devs[d]);
            if (ncclAllGather(b->d_send[d], b->d_recv[d], (size_t)slots * 6, ncclFloat, b->comms[d], (hipStream_t)bvh_ctx_stream(b->lane(d, 0))) != ncclSuccess) rc = BVH_E_INTERNAL;
        }
        if (ncclGroupEnd() != ncclSuccess) rc = BVH_E_INTERNAL;
        for (int d = 0; d < n_dev; ++d) { (void)hipSetDevice(b->devs[d]); (void)hipEventRecord(b->ev1[d], (hipStream_t)bvh_ctx_stream(b->lane(d, 0))); }
        for (int d = 0; d < n_dev && !rc; ++d) rc = bvh_ctx_synchronize(b->lane(d, 0));
        float worst = 0.f;
        for (int d = 0; d < n_dev && !rc; ++d) {
            float ms = 0.f; (void)hipSetDevice(b->devs[d]);
            if (hipEventElapsedTime(&ms, b->ev0[d], b->ev1[d]) == hipSuccess && ms > worst) worst = ms;
        }
        rep->allgather_us = worst * 1000.f;
    }
    if (!rc) {
        std::vector<float> host((size_t)slots * n_dev * 6);
        (void)hipSetDevice(b->devs[0]);
        if (hipMemcpy(host.data(), b->d_recv[0], host.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = BVH_E_INTERNAL;
        for (int m = 0; m < n_meshes && !rc; ++m) {
            const int d = m % n_dev, k = m / n_dev;
            for (int c = 0; c < 6; ++c) rep->root_aabbs[6 * m + c] = host[((size_t)d * slots + k) * 6 + c];
        }
    }
    rep->wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    rep->lanes_per_device = lanes;
    return rc;
}

// read-back of one kept tree (the C++ mirror's getData() of d_bvhNodes / d_primRefs): blocking
extern "C" int bvh_batch_download(bvh_batch* b, const bvh_batch_mesh* mesh, void* h_nodes, void* h_leaves) {
    if (!b || !mesh || !mesh->d_nodes) return BVH_E_INVALID_ARG;
    if (hipSetDevice(mesh->device) != hipSuccess) return BVH_E_INTERNAL;
    if (h_nodes && hipMemcpy(h_nodes, mesh->d_nodes, (size_t)mesh->n_nodes * 32, hipMemcpyDeviceToHost) != hipSuccess) return BVH_E_INTERNAL;
    if (h_leaves && mesh->d_leaves && hipMemcpy(h_leaves, mesh->d_leaves, (size_t)mesh->n_leaves * 28, hipMemcpyDeviceToHost) != hipSuccess) return BVH_E_INTERNAL;
    return 0;
}

extern "C" int bvh_batched_build(int n_dev, const int* devs, bvh_algo algo, const void* const* h_tris, const uint32_t* n_tris, int n_meshes,
                                 float* root_aabbs_out, float* build_ms_out) {
    if (n_dev <= 0 || !devs || !h_tris || !n_tris || n_meshes <= 0 || !root_aabbs_out) return BVH_E_INVALID_ARG;
    bvh_batch* b = nullptr;
    int rc = bvh_batch_create(n_dev, devs, &b); if (rc) return rc;
    bvh_batch_report rep{}; rep.root_aabbs = root_aabbs_out; rep.build_ms = build_ms_out;
    rc = bvh_batch_build(b, algo, h_tris, n_tris, n_meshes, &rep);
    bvh_batch_destroy(b);
    return rc;
}
