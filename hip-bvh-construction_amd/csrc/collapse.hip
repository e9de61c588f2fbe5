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
// created by launch k-1 (every level counts what it allocates in a word of its own, so the next launch derives its bounds from finished
// counts: no kernel in between), ids are allocated with one block-aggregated atomic per workgroup.  Numbering is allocation order (schedule dependent, as in the reference).  The host enqueues batches of
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

// Level bounds without a kernel in between (round 3; rounds 1-2 ran a one-thread snapshot kernel after every level: twice the launches).  A batch of levels
// shares state[0 .. COLLAPSE_MAX_BATCH): state[l] = wide nodes allocated BY level l of the batch, all zero when the batch starts.  Level l works on the
// ids [begin_l, end_l): begin_0 / end_0 come from the host (the root task, or what the previous batch's read-back says), begin_l = end_(l-1),
// end_l = begin_l + state[l-1] — every count a launch reads was finished by an earlier launch — and it allocates id = end_l + atomicAdd(&state[l], count).
__global__ void k_collapse_init(uint2* taskq, u32* state, u32 root, int with_root) {
    if (with_root && tid_x() == 0) taskq[0] = make_uint2(root, INV);                      // src/TwoPassLbvh.cpp:160-167
    for (int i = tid_x(); i < COLLAPSE_MAX_BATCH; i += bdim_x()) state[i] = 0u;
}

__global__ __launch_bounds__(CL_BLOCK) void k_collapse_level(const bvh2_node* __restrict__ nodes, const bvh_primref* __restrict__ leaves,
                                                             Wide4* __restrict__ wide, PrimNodeRec* __restrict__ prims, uint2* taskq,
                                                             u32* state, u32 n, int layout, u32 base_begin, u32 base_len, int level) {
    u32 begin = base_begin, len = base_len;
    for (int j = 0; j < level; ++j) { begin += len; len = state[j]; }        // (uniform; <= COLLAPSE_MAX_BATCH cached words)
    const u32 end = begin + len;
    if (len == 0u) return;
    u32* const alloc = state + level;
    const u32 ni = n - 1;
    __shared__ u32 s_base, s_count;
    auto box_of = [&](u32 c) -> Box { return (layout == 1 && c >= ni) ? box_load_u(&leaves[c - ni].aabb) : box_load(&nodes[c].aabb); };
    for (u32 g0 = begin + bid_x() * CL_BLOCK; g0 < end; g0 += nbid_x() * CL_BLOCK) {     // block-uniform
        const u32 g = g0 + tid_x();
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
        if (tid_x() == 0) s_count = 0;
        __syncthreads();
        u32 my_off = 0;
        if (n_int) my_off = atomicAdd(&s_count, n_int);
        __syncthreads();
        if (tid_x() == 0) s_base = s_count ? end + atomicAdd(alloc, s_count) : 0u;
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

// with_root: the collapse starts (the root task is queued); otherwise only the batch's counters are cleared for the next batch
void collapse_begin(hipStream_t s, uint2* d_taskq, u32* d_state, u32 root, bool with_root) {
    hipLaunchKernelGGL(k_collapse_init, dim3(1), dim3(64), 0, s, d_taskq, d_state, root, with_root ? 1 : 0);
}
// enqueue the `count` levels of one batch (a level without work returns at once); the batch's first level works on [base_begin, base_begin + base_len)
void collapse_enqueue(hipStream_t s, const void* d_nodes, const void* d_leaves, void* d_wide, void* d_prims, uint2* d_taskq,
                      u32* d_state, u32 base_begin, u32 base_len, int count, u32 n, int layout, const u32* expect) {
    u32 full = (n / 2 + CL_BLOCK - 1) / CL_BLOCK; if (full > 2048u) full = 2048u; if (full == 0) full = 1;
    KernelScope ks(s, "k_collapse_level");
    for (int level = 0; level < count; ++level) {
        // expect[level]: the tasks this level had in the previous collapse of a tree of this size (nullptr: unknown) — most levels of a wide tree are thin,
        // and a launch of hundreds of workgroups that find nothing costs more than the level's work; any grid is correct (grid-stride loop)
        u32 grid = full;
        if (expect) { const u32 want = (expect[level] + expect[level] / 4u + (u32)CL_BLOCK - 1u) / (u32)CL_BLOCK + 1u; if (want < grid) grid = want; }
        hipLaunchKernelGGL(k_collapse_level, dim3(grid), dim3(CL_BLOCK), 0, s, (const bvh2_node*)d_nodes, (const bvh_primref*)d_leaves,
                           (Wide4*)d_wide, (PrimNodeRec*)d_prims, d_taskq, d_state, n, layout, base_begin, base_len, level);
    }
}

// (kernels.hpp: touching one kernel of this translation unit makes the runtime load its code object — bvh_ctx_create does that for the build path's modules, so
// that a context's FIRST build does not pay for it: 0.3-0.7 ms per module on the MI355X, tools/cold_probe.py)
void warm_collapse() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_collapse_init)); }

} // namespace bvh
