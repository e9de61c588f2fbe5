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

void launch_to_lbvh_layout(hipStream_t s, const void* d_nodes, const void* d_leaves, uint32_t n, void* d_out) {
    hipLaunchKernelGGL(k_to_lbvh_layout, dim3((n + 255) / 256), dim3(256), 0, s, (const bvh2_node*)d_nodes, (const bvh_primref*)d_leaves, n, (bvh2_node*)d_out);
}

void launch_sah_cost(hipStream_t s, const void* d_nodes, const void* d_leaves, uint32_t root, uint32_t n, int layout, double* d_out) {
    hipMemsetAsync(d_out, 0, sizeof(double), s);
    const u32 blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_sah, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, s, (const bvh2_node*)d_nodes, (const bvh_primref*)d_leaves, root, n, layout, d_out);
}

} // namespace bvh
