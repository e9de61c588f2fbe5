// collapse.hip — BVH2 -> BVH4 collapse ("n-wide collapse" of BASELINE.json config 4).
//
// Replaces CollapseToWide4Bvh (reference src/TwoPassLbvhKernel.h:237-336 for the LBVH layout, src/Ploc++Kernel.h:364-465 for
// the PLOC layout) and its host set-up (src/TwoPassLbvh.cpp:154-183).  Semantics kept: wide node <- task
// {BVH2 node, parent wide node}; children = the BVH2 node's two children, then twice: the internal child with the largest
// area (strict >, first wins, zero-area internals never expand) is replaced in place by its left child and its right child
// is appended; internal children get fresh wide-node ids and tasks, leaves are recorded in PrimNode[leaf]; wide root = 0;
// boxes are stored for internal children only (leaf slots keep the reset box, as the reference's default-constructed node).
//
// The reference runs ONE launch in which every thread spins until its task appears (hangs unless all workgroups are
// co-resident).  Here the wide tree is produced level by level: launch k processes the wide nodes
// created by launch k-1 (a tiny snapshot kernel publishes the level bounds), ids are allocated with one block-aggregated
// atomic per workgroup.  Numbering is allocation order (schedule dependent, as in the reference).  The host enqueues batches of
// levels and reads the bounds back after each batch until a level created nothing: any depth works (a degenerate 1 M-leaf chain
// just takes more batches).  A one-launch variant (waves taking 64-task batches from a ticket and polling their queue words) was
// measured slower: 0.56 vs 0.30 ms at 262 k, 1.5 vs 1.2 ms at 10 M — a batch completes at the pace of its slowest task
// (tools/probes/collapse_persistent_queue.hip.txt).
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

struct alignas(128) Wide4 { bvh_aabb aabb[4]; u32 child[4]; u32 parent; u32 count; u32 pad[2]; };   // Bvh4Node, src/Common.h:560-566
struct PrimNodeRec { u32 prim, parent; };                                                           // PrimNode, src/Common.h:568-572
static_assert(sizeof(Wide4) == 128 && sizeof(PrimNodeRec) == 8, "reference layouts");

constexpr int CL_BLOCK = 256;

// state: [0] allocation counter (next free wide id), [1] / [2] first wide id of the current / the next level, [3] levels that had work
__global__ void k_collapse_init(uint2* taskq, u32* state, u32 root) {
    if (threadIdx.x == 0) { taskq[0] = make_uint2(root, INV); state[0] = 1; state[1] = 0; state[2] = 1; state[3] = 0; }   // src/TwoPassLbvh.cpp:160-167
}
// between two levels: the wide nodes the last level allocated are the next level
__global__ void k_collapse_snapshot(u32* state) { if (threadIdx.x == 0) { if (state[2] > state[1]) state[3] += 1; state[1] = state[2]; state[2] = state[0]; } }

__global__ __launch_bounds__(CL_BLOCK) void k_collapse_level(const bvh2_node* __restrict__ nodes, const bvh_primref* __restrict__ leaves,
                                                             Wide4* __restrict__ wide, PrimNodeRec* __restrict__ prims, uint2* taskq,
                                                             u32* state, u32 n, int layout) {
    const u32 begin = state[1], end = state[2];
    const u32 ni = n - 1;
    __shared__ u32 s_base, s_count;
    auto box_of = [&](u32 c) -> Box { return (layout == 1 && c >= ni) ? box_load_u(&leaves[c - ni].aabb) : box_load(&nodes[c].aabb); };
    for (u32 g0 = begin + blockIdx.x * CL_BLOCK; g0 < end; g0 += gridDim.x * CL_BLOCK) {     // block-uniform
        const u32 g = g0 + threadIdx.x;
        const bool have = g < end;
        u32 ci[4] = { INV, INV, INV, INV }; Box cb[4]; u32 cc = 0, parent = INV, n_int = 0;
        if (have) {
            const uint2 task = taskq[g];
            parent = task.y;
            const u32 l = nodes[task.x].left, r = nodes[task.x].right;
            ci[0] = l; ci[1] = r; cb[0] = box_of(l); cb[1] = box_of(r); cc = 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {                                                      // :270-296
                float best = 0.0f; u32 pos = INV;
                for (u32 k = 0; k < cc; ++k)
                    if (ci[k] < ni) { const float a = box_area(box_load(&nodes[ci[k]].aabb)); if (a > best) { pos = k; best = a; } }
                if (pos == INV) break;
                const u32 ex = ci[pos];
                const u32 el = nodes[ex].left, er = nodes[ex].right;
#pragma unroll
                for (int k = 0; k < 4; ++k) if ((u32)k == pos) { ci[k] = el; cb[k] = box_of(el); }
#pragma unroll
                for (int k = 0; k < 4; ++k) if ((u32)k == cc) { ci[k] = er; cb[k] = box_of(er); }
                ++cc;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) if ((u32)k < cc && ci[k] < ni) ++n_int;
        }
        // block-aggregated allocation of the internal children's wide ids
        __syncthreads();
        if (threadIdx.x == 0) s_count = 0;
        __syncthreads();
        u32 my_off = 0;
        if (n_int) my_off = atomicAdd(&s_count, n_int);
        __syncthreads();
        if (threadIdx.x == 0) s_base = s_count ? atomicAdd(&state[0], s_count) : 0u;
        __syncthreads();
        if (have) {
            Wide4 w;
            u32 next = s_base + my_off;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w.child[k] = INV;
                w.aabb[k].min = { FMAX, FMAX, FMAX }; w.aabb[k].max = { -FMAX, -FMAX, -FMAX };
                if ((u32)k < cc) {
                    if (ci[k] < ni) {                                                          // :312-318
                        w.child[k] = next; box_store(&w.aabb[k], cb[k]);
                        taskq[next] = make_uint2(ci[k], g); ++next;
                    } else {                                                                   // :319-324
                        w.child[k] = ci[k];
                        const u32 leaf = ci[k] - ni;
                        prims[leaf].parent = g;
                        prims[leaf].prim = (layout == 1) ? leaves[leaf].prim_idx : nodes[ci[k]].left;
                    }
                }
            }
            w.parent = parent; w.count = cc; w.pad[0] = 0; w.pad[1] = 0;
            wide[g] = w;
        }
    }
}

void collapse_begin(hipStream_t s, uint2* d_taskq, u32* d_state, u32 root) {
    hipLaunchKernelGGL(k_collapse_init, dim3(1), dim3(64), 0, s, d_taskq, d_state, root);
}
// enqueue `count` more levels (a level without work returns at once); `first`: the very first level follows collapse_begin directly
void collapse_enqueue(hipStream_t s, const void* d_nodes, const void* d_leaves, void* d_wide, void* d_prims, uint2* d_taskq,
                      u32* d_state, bool first, int count, u32 n, int layout) {
    u32 grid = (n / 2 + CL_BLOCK - 1) / CL_BLOCK; if (grid > 2048u) grid = 2048u; if (grid == 0) grid = 1;
    KernelScope ks(s, "k_collapse_level");
    for (int level = 0; level < count; ++level) {
        if (!(first && level == 0)) hipLaunchKernelGGL(k_collapse_snapshot, dim3(1), dim3(64), 0, s, d_state);
        hipLaunchKernelGGL(k_collapse_level, dim3(grid), dim3(CL_BLOCK), 0, s, (const bvh2_node*)d_nodes, (const bvh_primref*)d_leaves,
                           (Wide4*)d_wide, (PrimNodeRec*)d_prims, d_taskq, d_state, n, layout);
    }
}

} // namespace bvh
