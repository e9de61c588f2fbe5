// hploc.hip — stage B for HPLOC on gfx950 (wave64).
//
// Replaces SetupClusters + HPloc of the reference (src/HplocKernel.h:39-56, :66-81 findParent, :83-117
// findNearestNeighbours, :126-190 mergeClusters, :192-218 load/storeIndices, :220-255 plocMerge, :257-315 HPloc; host
// src/Hploc.cpp:83-121).  Output: Bvh2Node[n-1] (root = node 0) + PrimRef[n] leaves in Morton order.  The tree topology is
// the reference's (threshold 16, <= 32 clusters per merge step, search radius 8, mutual nearest neighbours under the
// {area bits, index} order); node numbering follows allocation order and is schedule dependent, as in the reference.
//
// The reference runs one wave32 per 32 leaves and keeps the 32-entry work list in LDS with implicit lock-step and a
// conflicting-store compaction.  Here one wave64 walks 64 leaves; the work list of a merge task
// lives in registers of lanes 0..31 and moves with cross-lane operations only (ds_bpermute / ds_permute / DPP) — no LDS
// allocation, no barriers, no reliance on store conflict order.
//
// Hand-off between waves (possibly on different XCDs): cluster ids (cidx) and internal-node boxes written during the
// launch are agent-scope write-through stores, read back with agent-scope loads; the wave drains its stores before the
// agent-scope atomic exchange on parent[] that hands the finished range to the sibling's walker.
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

constexpr int HP_BLOCK = 64;       // one wave per workgroup
constexpr u32 HP_HALF = 16;        // WarpSize/2 of the reference's wave32 (src/HplocKernel.h:195,238)
constexpr int HP_RADIUS = 8;       // PlocRadius, src/Common.h:595

__global__ __launch_bounds__(256) void k_setup_clusters(const bvh_aabb* __restrict__ boxes, const u32* __restrict__ svals,
                                                        bvh_primref* __restrict__ leaves, u32* __restrict__ cidx,
                                                        u32* __restrict__ parent, u32 n) {   // SetupClusters :39-56
    const u32 g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const u32 prim = svals[g];
    const Box b = box_load(boxes + prim);
    float* f = reinterpret_cast<float*>(leaves + g);
    reinterpret_cast<u32*>(f)[0] = prim;
    f[1] = b.lx; f[2] = b.ly; f[3] = b.lz; f[4] = b.hx; f[5] = b.hy; f[6] = b.hz;
    cidx[g] = g + (n - 1);
    if (parent) parent[g] = INV;
}

__device__ __forceinline__ Box shfl_box(const Box& b, int src) {
    return { __shfl(b.lx, src), __shfl(b.ly, src), __shfl(b.lz, src), __shfl(b.hx, src), __shfl(b.hy, src), __shfl(b.hz, src) };
}
__device__ __forceinline__ Box shfl_down_box(const Box& b, int d) {
    return { __shfl_down(b.lx, d), __shfl_down(b.ly, d), __shfl_down(b.lz, d), __shfl_down(b.hx, d), __shfl_down(b.hy, d), __shfl_down(b.hz, d) };
}
// push semantics: lane l's value lands in lane dst(l)
__device__ __forceinline__ u32 push_u32(int dst, u32 v) { return (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)v); }
__device__ __forceinline__ float push_f32(int dst, float v) { return __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(v))); }

// One plocMerge (:220-255) for the range [tL, tR) split at tS, executed by the whole wave; slots = lanes 0..31.
__device__ __forceinline__ void merge_task(u32 tL, u32 tS, u32 tR, bool final_, const bvh_primref* __restrict__ leaves,
                                           bvh2_node* nodes, u32* cidx, u32* counter, u32 ni, int lane, u64 lt) {
    // loadIndices (:192-206): the first min(len,16) ids of each child range; valid ones form a prefix
    const u32 lenL = min(tS - tL, HP_HALF), lenR = min(tR - tS, HP_HALF);
    u32 id = INV;
    if (lane < 16) { if ((u32)lane < lenL) id = ld_agent(cidx + tL + lane); }
    else if (lane < 32) { if ((u32)(lane - 16) < lenR) id = ld_agent(cidx + tS + (lane - 16)); }
    const u64 vb = __ballot(id != INV);
    const u32 nl = (u32)__popcll(vb & 0xFFFFull), nr = (u32)__popcll(vb & 0xFFFF0000ull);
    const u32 loaded = nl + nr;
    u32 cnt = loaded;
    {   // left-pack: slot s < nl <- lane s ; slot s in [nl, cnt) <- lane 16 + (s - nl)
        const int src = ((u32)lane < nl) ? lane : (int)(16 + (u32)lane - nl);
        const u32 t = (u32)__shfl((int)id, src & 63);
        id = ((u32)lane < cnt) ? t : INV;
    }
    Box b = box_empty();
    if (id != INV) b = (id >= ni) ? box_load(&leaves[id - ni].aabb) : node_box_agent(nodes + id);   // :242-246
    const u32 threshold = final_ ? 1u : HP_HALF;
    while (cnt > threshold) {
        // findNearestNeighbours (:83-117): key = {area bits, neighbour slot}; both directions evaluated from one area
        u64 best = ~0ull;
#pragma unroll
        for (int r = 1; r <= HP_RADIUS; ++r) {
            const Box nb = shfl_down_box(b, r);
            const u32 ab = __float_as_uint(box_area(box_union(nb, b)));
            const u32 ab_left = (u32)__shfl_up((int)ab, r);
            if ((u32)(lane + r) < cnt) { const u64 k = ((u64)ab << 32) | (u32)(lane + r); best = k < best ? k : best; }
            if (lane >= r && (u32)lane < cnt) { const u64 k = ((u64)ab_left << 32) | (u32)(lane - r); best = k < best ? k : best; }
        }
        // mergeClusters (:126-190)
        const int nbr = (int)((u32)best & 63u);
        const u32 nbr_of_nbr = (u32)__shfl((int)(u32)best, nbr);
        const bool in = (u32)lane < cnt;
        const bool mutual = in && nbr_of_nbr == (u32)lane;
        const bool merge = mutual && lane < nbr;
        const bool absorbed = mutual && lane > nbr;
        const u64 mm = __ballot(merge);
        const u32 total = (u32)__popcll(mm);
        u32 base = 0;
        if (lane == 0) base = __hip_atomic_fetch_add(counter, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // :163
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        const u32 new_node = ni - base - total + (u32)__popcll(mm & lt);                                          // :165-167
        const u32 id_nb = (u32)__shfl((int)id, nbr);
        const Box bn = shfl_box(b, nbr);
        if (merge) {
            b = box_union(b, bn);
            node_store_agent(nodes + new_node, id, id_nb, b);
            id = new_node;
        }
        // compaction: survivors and merged clusters keep their order (:176-187, as "valid lanes write to their rank")
        const bool keep = in && !absorbed;
        const u64 km = __ballot(keep);
        const u32 newcnt = (u32)__popcll(km);
        const int dst = keep ? (int)__popcll(km & lt) : 63;    // lane 63 is never a slot: harmless sink
        id = push_u32(dst, id);
        b.lx = push_f32(dst, b.lx); b.ly = push_f32(dst, b.ly); b.lz = push_f32(dst, b.lz);
        b.hx = push_f32(dst, b.hx); b.hy = push_f32(dst, b.hy); b.hz = push_f32(dst, b.hz);
        if ((u32)lane >= newcnt) id = INV;
        cnt = newcnt;
    }
    if ((u32)lane < loaded) st_agent(cidx + tL + lane, id);      // storeIndices (:208-218)
}

__global__ __launch_bounds__(HP_BLOCK) void k_hploc(const bvh_primref* __restrict__ leaves, const u32* __restrict__ skeys,
                                                    bvh2_node* nodes, u32* cidx, u32* parent, u32* counter, u32 n) {
    const int lane = threadIdx.x;
    const u64 lt = (1ull << lane) - 1ull;
    const u32 g = blockIdx.x * HP_BLOCK + (u32)lane;
    const u32 ni = n - 1;
    u32 L = g, R = g;                       // inclusive range
    bool active = g < n;                    // covers every leaf (the reference under-launches when (n-1)%32==0, App. B)
    while (__ballot(active)) {
        u32 split = INV;
        if (active) {
            // findParent (:66-81) with inclusive R: hand over to the split at R (as left child) or at L-1 (as right child)
            bool to_right;
            if (L == 0) to_right = true;
            else if (R == ni) to_right = false;
            else to_right = (aug_key(skeys, R) ^ aug_key(skeys, R + 1)) < (aug_key(skeys, L - 1) ^ aug_key(skeys, L));
            drain_stores();                 // the wave's node / cidx stores are in memory before the range is handed over
            u32 prev;
            if (to_right) {
                prev = __hip_atomic_exchange(parent + R, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // :278
                if (prev != INV) { split = R + 1; R = prev; }
            } else {
                prev = __hip_atomic_exchange(parent + (L - 1), R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // :288
                if (prev != INV) { split = L; L = prev; }
            }
            if (prev == INV) active = false;
        }
        compiler_fence();
        const u32 size = R - L + 1;
        const bool final_ = active && size == n;
        u64 todo = __ballot((active && size > HP_HALF) || final_);                                           // :305
        while (todo) {
            const int owner = __ffsll((unsigned long long)todo) - 1;
            const u32 tL = (u32)__shfl((int)L, owner), tS = (u32)__shfl((int)split, owner), tR = (u32)__shfl((int)R, owner) + 1u;
            const bool tF = __shfl((int)final_, owner) != 0;
            merge_task(tL, tS, tR, tF, leaves, nodes, cidx, counter, ni, lane, lt);
            todo &= todo - 1;
        }
    }
}

void launch_setup_clusters(hipStream_t s, const void* d_boxes, const uint32_t* d_svals, uint32_t n, void* d_leaves,
                           uint32_t* d_cluster_idx, uint32_t* d_parent) {
    { KernelScope ks(s, "k_setup_clusters"); hipLaunchKernelGGL(k_setup_clusters, dim3((n + 255) / 256), dim3(256), 0, s, (const bvh_aabb*)d_boxes, d_svals,
                       (bvh_primref*)d_leaves, d_cluster_idx, d_parent, n); }
}

void launch_hploc(hipStream_t s, const void* d_boxes, const uint32_t* d_skeys, const uint32_t* d_svals, uint32_t n,
                  void* d_nodes, void* d_leaves, uint32_t* d_cluster_idx, uint32_t* d_parent, uint32_t* d_counter) {
    hipMemsetAsync(d_counter, 0, sizeof(u32), s);
    launch_setup_clusters(s, d_boxes, d_svals, n, d_leaves, d_cluster_idx, d_parent);
    { KernelScope ks(s, "k_hploc"); hipLaunchKernelGGL(k_hploc, dim3((n + HP_BLOCK - 1) / HP_BLOCK), dim3(HP_BLOCK), 0, s, (const bvh_primref*)d_leaves, d_skeys,
                       (bvh2_node*)d_nodes, d_cluster_idx, d_parent, d_counter, n); }
}

} // namespace bvh
