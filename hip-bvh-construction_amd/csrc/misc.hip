// misc.hip — consumers' helpers: PLOC->LBVH layout adapter and a device-side SAH cost reduction.
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

// PLOC layout (nodes[n-1] + PrimRef leaves[n]) -> LBVH layout (one array of 2n-1, leaf i at n-1+i with
// {left = primIdx, right = INVALID, aabb}, reference src/TwoPassLbvhKernel.h:177-184).  Child indices already follow the
// ">= n-1 is a leaf" convention (src/Ploc++Kernel.h:47), so internal nodes are copied verbatim.
__global__ __launch_bounds__(256) void k_to_lbvh_layout(const bvh2_node* __restrict__ nodes, const bvh_primref* __restrict__ leaves,
                                                        u32 n, bvh2_node* __restrict__ out) {
    const u32 g = blockIdx.x * 256 + threadIdx.x;
    const u32 ni = n - 1;
    if (g < ni) {
        const uint4* src = reinterpret_cast<const uint4*>(nodes + g);
        uint4* dst = reinterpret_cast<uint4*>(out + g);
        dst[0] = src[0]; dst[1] = src[1];
    }
    if (g < n) {
        bvh2_node* o = out + ni + g;
        o->left = leaves[g].prim_idx; o->right = INV;
        box_store(&o->aabb, box_load_u(&leaves[g].aabb));
    }
}

// BVH2 SAH cost, formula of Utility::calculateLbvhCost (reference src/Utility.cpp:317-349):
//   1 + sum over internal nodes (area(left) + area(right)) / area(root) + sum over leaves area(leaf) / area(root)
// areas in f32 exactly as the reference computes them, accumulation in f64.
__global__ __launch_bounds__(256) void k_sah(const bvh2_node* __restrict__ nodes, const bvh_primref* __restrict__ leaves,
                                             u32 root, u32 n, int layout, double* __restrict__ out) {
    const u32 ni = n - 1;
    auto area_of = [&](u32 c) -> float {
        return (layout == 1 && c >= ni) ? box_area(box_load_u(&leaves[c - ni].aabb)) : box_area(box_load(&nodes[c].aabb));
    };
    const double ra = (double)area_of(root);
    double acc = 0.0;
    const u32 stride = gridDim.x * 256;
    for (u32 g = blockIdx.x * 256 + threadIdx.x; g < n; g += stride) {
        if (g < ni) {
            const u32 l = nodes[g].left, r = nodes[g].right;
            if (l != INV) acc += (double)area_of(l) / ra;
            if (r != INV) acc += (double)area_of(r) / ra;
        }
        if (layout == 1) acc += (double)box_area(box_load_u(&leaves[g].aabb)) / ra;
        else if (nodes[ni + g].left != INV) acc += (double)box_area(box_load(&nodes[ni + g].aabb)) / ra;
    }
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) acc += __shfl_xor(acc, m);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0] + red[1] + red[2] + red[3];
        if (blockIdx.x == 0) t += 1.0;
        atomicAdd(out, t);
    }
}

// BVH4 cost, formula of Utility::calculatebvh4Cost (reference src/Utility.cpp:351-396; what m_cost means after X::build, e.g.
// src/TwoPassLbvh.cpp:196): 1 + sum over wide nodes and their INTERNAL children (child < n-1) of area(child box) / area(root) + sum over
// the n primitives of area(prim box) / area(root); root box = union of the wide root's child boxes (leaf slots hold the reset box and do
// not contribute, exactly as the reference's grow() of a default-constructed Aabb).  Terms in f32 as the reference computes them
// (area * rootInvArea), accumulation in f64 (the reference accumulates in f32 in node-index order, so its last bits depend on its
// schedule-dependent numbering).
struct alignas(128) Wide4Rec { bvh_aabb aabb[4]; u32 child[4]; u32 parent; u32 count; u32 pad[2]; };     // Bvh4Node, src/Common.h:560-566
__global__ __launch_bounds__(256) void k_bvh4_cost(const Wide4Rec* __restrict__ wide, u32 n_wide, const uint2* __restrict__ prims,
                                                   const bvh_aabb* __restrict__ prim_boxes, u32 n, double* __restrict__ out) {
    const u32 ni = n - 1;
    Box rb = box_empty();
#pragma unroll
    for (int k = 0; k < 4; ++k) if (wide[0].child[k] != INV) rb = box_union(rb, box_load(&wide[0].aabb[k]));
    const float inv = 1.0f / box_area(rb);
    double acc = 0.0;
    const u32 stride = gridDim.x * 256;
    for (u32 g = blockIdx.x * 256 + threadIdx.x; g < n; g += stride) {
        if (g < n_wide) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const u32 c = wide[g].child[k]; if (c != INV && c < ni) acc += (double)(box_area(box_load(&wide[g].aabb[k])) * inv); }
        }
        acc += (double)(box_area(box_load(prim_boxes + prims[g].x)) * inv);
    }
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) acc += __shfl_xor(acc, m);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0] + red[1] + red[2] + red[3];
        if (blockIdx.x == 0) t += 1.0;
        atomicAdd(out, t);
    }
}

// Order-independent 64-bit checksum of a build result: sum (mod 2^64) over every node / leaf record of a hash of {array tag, index,
// the record's dwords}, plus the root index.  Two builds have the same checksum iff (up to hash collisions) their arrays are byte-identical;
// integer addition commutes, so the value does not depend on the schedule.  Mirrored in numpy by the test harness (checksum_host).
__device__ __forceinline__ u64 ck_mix(u64 h, u32 w) { h = (h ^ (u64)w) * 0xff51afd7ed558ccdull; return h ^ (h >> 32); }
__global__ __launch_bounds__(256) void k_checksum(const u32* __restrict__ nodes, u32 n_nodes, const u32* __restrict__ leaves, u32 n_leaves, u32 root,
                                                  unsigned long long* __restrict__ out) {
    u64 acc = 0ull;
    const u32 stride = gridDim.x * 256;
    const u32 total = n_nodes > n_leaves ? n_nodes : n_leaves;
    for (u32 g = blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
        if (g < n_nodes) {
            u64 h = 0x9E3779B97F4A7C15ull * ((u64)g + 1ull) + 1ull;
            const uint4 a = reinterpret_cast<const uint4*>(nodes)[2 * (size_t)g], b = reinterpret_cast<const uint4*>(nodes)[2 * (size_t)g + 1];
            h = ck_mix(h, a.x); h = ck_mix(h, a.y); h = ck_mix(h, a.z); h = ck_mix(h, a.w); h = ck_mix(h, b.x); h = ck_mix(h, b.y); h = ck_mix(h, b.z); h = ck_mix(h, b.w);
            acc += h;
        }
        if (leaves && g < n_leaves) {
            u64 h = 0x9E3779B97F4A7C15ull * ((u64)g + 1ull) + 2ull;
#pragma unroll
            for (int k = 0; k < 7; ++k) h = ck_mix(h, leaves[(size_t)g * 7 + k]);
            acc += h;
        }
    }
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out, (unsigned long long)ck_mix(3ull, root));
}

void launch_bvh4_cost(hipStream_t s, const void* d_wide, uint32_t n_wide, const void* d_prims, const void* d_prim_boxes, uint32_t n, double* d_out) {
    hipMemsetAsync(d_out, 0, sizeof(double), s);
    const u32 blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_bvh4_cost, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, s, (const Wide4Rec*)d_wide, n_wide, (const uint2*)d_prims, (const bvh_aabb*)d_prim_boxes, n, d_out);
}
void launch_checksum(hipStream_t s, const void* d_nodes, uint32_t n_nodes, const void* d_leaves, uint32_t n_leaves, uint32_t root, uint64_t* d_out) {
    hipMemsetAsync(d_out, 0, sizeof(uint64_t), s);
    const u32 total = n_nodes > n_leaves ? n_nodes : n_leaves;
    const u32 blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_checksum, dim3(blocks < 2048u ? blocks : 2048u), dim3(256), 0, s, (const u32*)d_nodes, n_nodes, (const u32*)d_leaves, d_leaves ? n_leaves : 0u, root, (unsigned long long*)d_out);
}

// Small read-backs (single-pass LBVH root index, PLOC++ iteration state, the collapse's level counts): one workgroup writes every word as a PAIR {v, ~v} straight into
// the context's pinned host words (host memory is device-accessible), and the host polls until every pair is consistent (api.hip wait_readback).  Round 3 copied the
// words with a 4-byte hipMemcpyAsync and polled for "not the sentinel any more": that copy is performed BYTE-WISE — under the batched builder's three concurrent streams
// the host read 0xff000972 and 0xffff09e9 (two and three bytes of a root index landed, the rest still the sentinel) and turned them into addresses (round 4,
// tools/probes/batch_stress.py).  A pair can only be consistent when every byte of both words is final.
__global__ __launch_bounds__(64) void k_readback(const u32* __restrict__ a, u32 na, const u32* __restrict__ b, u32 nb, u32* out) {
    for (u32 i = threadIdx.x; i < na + nb; i += 64) {
        const u32 v = i < na ? a[i] : b[i - na];
        __hip_atomic_store(out + 2 * i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(out + 2 * i + 1, ~v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
void launch_readback(hipStream_t s, const uint32_t* d_a, uint32_t na, const uint32_t* d_b, uint32_t nb, uint32_t* pinned_pairs) {
    hipLaunchKernelGGL(k_readback, dim3(1), dim3(64), 0, s, d_a, na, d_b, nb, pinned_pairs);
}

// device-to-device copy as a kernel (bvh_dev_copy).  hipMemcpyAsync(DeviceToDevice) of a few bytes takes a host-side path in the runtime (a CPU memcpy through the
// BAR mapping) that crashed when several host threads of the batched builder copied while another thread re-allocated its arena (round 4: SIGSEGV inside
// hipMemcpyAsync under tools/probes/batch_stress.py); a kernel only ever touches the memory from the device, on the caller's stream.
template <typename T>
__global__ __launch_bounds__(256) void k_copy(T* __restrict__ dst, const T* __restrict__ src, size_t count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
void launch_copy_bytes(hipStream_t s, void* d_dst, const void* d_src, size_t bytes) {
    if (!bytes) return;
    const uintptr_t a = (uintptr_t)d_dst | (uintptr_t)d_src | (uintptr_t)bytes;
    auto grid = [](size_t count) { const size_t b = (count + 255) / 256; return dim3((unsigned)(b < 4096 ? b : 4096)); };
    if ((a & 15u) == 0) hipLaunchKernelGGL(k_copy<uint4>, grid(bytes / 16), dim3(256), 0, s, (uint4*)d_dst, (const uint4*)d_src, bytes / 16);
    else if ((a & 3u) == 0) hipLaunchKernelGGL(k_copy<u32>, grid(bytes / 4), dim3(256), 0, s, (u32*)d_dst, (const u32*)d_src, bytes / 4);
    else hipLaunchKernelGGL(k_copy<unsigned char>, grid(bytes), dim3(256), 0, s, (unsigned char*)d_dst, (const unsigned char*)d_src, bytes);
}

void launch_to_lbvh_layout(hipStream_t s, const void* d_nodes, const void* d_leaves, uint32_t n, void* d_out) {
    hipLaunchKernelGGL(k_to_lbvh_layout, dim3((n + 255) / 256), dim3(256), 0, s, (const bvh2_node*)d_nodes, (const bvh_primref*)d_leaves, n, (bvh2_node*)d_out);
}

void launch_sah_cost(hipStream_t s, const void* d_nodes, const void* d_leaves, uint32_t root, uint32_t n, int layout, double* d_out) {
    hipMemsetAsync(d_out, 0, sizeof(double), s);
    const u32 blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_sah, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, s, (const bvh2_node*)d_nodes, (const bvh_primref*)d_leaves, root, n, layout, d_out);
}

// (kernels.hpp: touching one kernel of this translation unit makes the runtime load its code object — bvh_ctx_create does that for the build path's modules, so
// that a context's FIRST build does not pay for it: 0.3-0.7 ms per module on the MI355X, tools/cold_probe.py)
void warm_misc() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_readback)); }

} // namespace bvh
