// batched.hip — scene-level multi-GPU shard inside ONE process (BASELINE.json config 5 through the C ABI).
//
// API name follows the reference's BatchedBvhBuilder::build(Context&, std::vector<BatchedBuildInput>&)
// (src/BatchedBuilder.h:12-31; its kernel builds many <=32-primitive trees on one GPU and does not compile — SURVEY.md
// Appendix B — so only the name and the "many independent meshes in, one result per mesh out" shape are kept).
// mesh m is built on devs[m % n_dev] by the ordinary single-GPU path (one host thread + one bvh_ctx per device, no peer
// traffic, no tree is ever split); afterwards ONE ncclAllGather of the per-device root-AABB slots (RCCL, xGMI) leaves the TLAS
// input on every device, and device devs[0]'s copy is returned to the host.  24 bytes per mesh: latency-only collective.
// (The multi-process flavour — one rank per GPU over torch.distributed — lives in the Python harness: batched.py, bench.py.)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <thread>
#include <vector>
#include <atomic>
#include "bvh_mi355x.h"

extern "C" int bvh_batched_build(int n_dev, const int* devs, bvh_algo algo, const void* const* h_tris, const uint32_t* n_tris, int n_meshes,
                                 float* root_aabbs_out, float* build_ms_out) {
    if (n_dev <= 0 || !devs || !h_tris || !n_tris || n_meshes <= 0 || !root_aabbs_out) return BVH_E_INVALID_ARG;
    const int slots = (n_meshes + n_dev - 1) / n_dev;
    std::vector<bvh_ctx*> ctx(n_dev, nullptr);
    std::vector<float*> d_send(n_dev, nullptr), d_recv(n_dev, nullptr);
    std::atomic<int> err{0};
    auto fail = [&](int code) { int z = 0; err.compare_exchange_strong(z, code); };
    // ---- builds: one host thread per device
    std::vector<std::thread> th;
    for (int d = 0; d < n_dev; ++d) th.emplace_back([&, d]() {
        int rc = bvh_ctx_create(devs[d], &ctx[d]); if (rc) return fail(rc);
        bvh_ctx_set_profiling(ctx[d], build_ms_out ? 1 : 0);
        if (hipSetDevice(devs[d]) != hipSuccess) return fail(BVH_E_INTERNAL);
        if (hipMalloc(&d_send[d], (size_t)slots * 6 * sizeof(float)) != hipSuccess || hipMalloc(&d_recv[d], (size_t)slots * n_dev * 6 * sizeof(float)) != hipSuccess) return fail(BVH_E_INTERNAL);
        if (hipMemset(d_send[d], 0, (size_t)slots * 6 * sizeof(float)) != hipSuccess) return fail(BVH_E_INTERNAL);
        int k = 0;
        for (int m = d; m < n_meshes; m += n_dev, ++k) {
            bvh_result r; bvh_timings t;
            rc = bvh_build(ctx[d], algo, h_tris[m], n_tris[m], 0, &r, &t); if (rc) return fail(rc);
            if (build_ms_out) build_ms_out[m] = t.ms_total;
            // root AABB = nodes[root].aabb (24 bytes at offset 8 of the 32-byte node)
            rc = bvh_dev_copy(ctx[d], d_send[d] + 6 * k, (const char*)r.d_nodes + 32 * (size_t)r.root + 8, 24); if (rc) return fail(rc);
        }
        rc = bvh_ctx_synchronize(ctx[d]); if (rc) return fail(rc);
    });
    for (auto& t : th) t.join();
    int rc = err.load();
    // ---- the only collective: all-gather of the root-AABB slots
    std::vector<ncclComm_t> comms(n_dev);
    bool comm_ok = false;
    if (!rc) {
        if (ncclCommInitAll(comms.data(), n_dev, devs) != ncclSuccess) rc = BVH_E_INTERNAL; else comm_ok = true;
    }
    if (!rc) {
        ncclGroupStart();
        for (int d = 0; d < n_dev; ++d) {
            (void)hipSetDevice(devs[d]);
            if (ncclAllGather(d_send[d], d_recv[d], (size_t)slots * 6, ncclFloat, comms[d], (hipStream_t)bvh_ctx_stream(ctx[d])) != ncclSuccess) rc = BVH_E_INTERNAL;
        }
        ncclGroupEnd();
        for (int d = 0; d < n_dev && !rc; ++d) rc = bvh_ctx_synchronize(ctx[d]);
    }
    if (!rc) {
        std::vector<float> host((size_t)slots * n_dev * 6);
        (void)hipSetDevice(devs[0]);
        if (hipMemcpy(host.data(), d_recv[0], host.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = BVH_E_INTERNAL;
        for (int m = 0; m < n_meshes && !rc; ++m) {
            const int d = m % n_dev, k = m / n_dev;
            for (int c = 0; c < 6; ++c) root_aabbs_out[6 * m + c] = host[((size_t)d * slots + k) * 6 + c];
        }
    }
    for (int d = 0; d < n_dev; ++d) {
        if (comm_ok) ncclCommDestroy(comms[d]);
        (void)hipSetDevice(devs[d]);
        if (d_send[d]) (void)hipFree(d_send[d]);
        if (d_recv[d]) (void)hipFree(d_recv[d]);
        if (ctx[d]) bvh_ctx_destroy(ctx[d]);
    }
    return rc;
}
