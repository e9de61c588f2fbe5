// lbvh.hip — stage B for the two LBVH builders on gfx950.
//
// Single pass (Apetrei 2014): replaces InitBvhNodes + BvhBuildAndFit, reference src/SinglePassLbvhKernel.h:27-126.
// Two pass   (Karras 2012)  : replaces InitBvhNodesPrimRef + BvhBuild + FitBvhNodes, src/TwoPassLbvhKernel.h:42-235.
// Output: Bvh2Node[2n-1] in the reference's LBVH layout, byte-identical to the reference's result (node index = split
// position for single pass; node i = the Karras node of position i for two pass).
//
// Inter-workgroup hand-off (MI355X: 8 XCDs, private L2s, L1s never refreshed by other CUs): every node field that another
// walker may read inside the launch is written with agent-scope write-through stores and read with agent-scope loads; the
// producer drains its stores (s_waitcnt vmcnt(0)) before the agent-scope RMW that publishes it.  No __threadfence():
// the reference's two fences per level (src/SinglePassLbvhKernel.h:107,124) would each write back / invalidate a cache.
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

constexpr int LBVH_BLOCK = 256;
constexpr u64 SLOT_EMPTY = ~0ull;

// ------------------------------------------------------------------------------------------------------------------
// Single pass.  One walker per leaf.  A finished node covering sorted positions [i,j) chooses its parent exactly as
// findParent (:64-86): parent = j-1 (as left child) if i==0 or (j!=n and delta(j-1,j) < delta(i-1,i)), else i-1 (as right
// child), delta = xor of the 64-bit {key,position} words.  Instead of the reference's {child link store, span store,
// counter atomicAdd, re-load} the two children of a node meet through ONE 64-bit atomic exchange on slots[parent]:
// the first arriver leaves {its node index, the far end of its range} and retires; the second arriver receives it, so it
// knows both children and the parent's full range, reads the sibling's box, and writes the parent node once (32 B).
// ------------------------------------------------------------------------------------------------------------------
template <typename K>
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_single(const bvh_aabb* __restrict__ boxes, const K* __restrict__ skeys,
                                                            const u32* __restrict__ svals, bvh2_node* nodes, u64* slots,
                                                            u32* root_out, u32 n) {
    const u32 g = blockIdx.x * LBVH_BLOCK + threadIdx.x;
    if (g >= n) return;
    const u32 ni = n - 1;
    const u32 prim = svals[g];
    Box box = box_load(boxes + prim);                       // = bounds of Triangle[prim] (:44), computed once in stage E
    node_store_agent(nodes + ni + g, prim, INV, box);       // leaf record {left = primIdx, right = INVALID} (:36-45)
    u32 i = g, j = g + 1, cur = ni + g;
    while (true) {
        if (i == 0 && j == n) { *root_out = cur; break; }   // :73 -> root (:116-120)
        bool as_left;
        if (i == 0) as_left = true;
        else if (j == n) as_left = false;
        else as_left = closer(skeys, j - 1, i - 1);
        const u32 p = as_left ? j - 1 : i - 1;
        const u64 mine = ((u64)cur << 32) | (as_left ? i : j);
        drain_stores();                                     // my node is in memory before anybody can learn its index
        const u64 other = __hip_atomic_exchange(slots + p, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (other == SLOT_EMPTY) break;                     // first arriver retires (atomicAdd(...) > 0 fails, :103)
        compiler_fence();
        const u32 sib = (u32)(other >> 32), far = (u32)other;
        box = box_union(box, node_box_agent(nodes + sib));  // merge(left.aabb, right.aabb) (:112) — min/max commute
        node_store_agent(nodes + p, as_left ? cur : sib, as_left ? sib : cur, box);
        if (as_left) j = far; else i = far;
        cur = p;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Two pass.  k_karras: node i's range and split from the sorted keys (:42-130), child links + parent pointers, leaf
// records.  k_refit: bottom-up boxes, second arriver at flags[parent] continues (:217-235).
// ------------------------------------------------------------------------------------------------------------------
// The block's key window [g0 - 256, g0 + 512] is staged in LDS: the exponential / binary searches of determineRange and
// findSplit probe it instead of paying an L2 round trip per probe; only ranges reaching beyond the window read global memory.
template <typename K>
__global__ __launch_bounds__(LBVH_BLOCK) void k_karras(const bvh_aabb* __restrict__ boxes, const K* __restrict__ k,
                                                       const u32* __restrict__ svals, bvh2_node* __restrict__ nodes,
                                                       u32* __restrict__ parent, u32 n) {
    __shared__ K s_keys[LBVH_BLOCK * 3 + 1];
    const int g0 = (int)(blockIdx.x * LBVH_BLOCK), w0 = g0 - LBVH_BLOCK;
    for (int t = threadIdx.x; t < LBVH_BLOCK * 3 + 1; t += LBVH_BLOCK) { const int j = w0 + t; s_keys[t] = (j >= 0 && j < (int)n) ? k[j] : (K)0; }
    __syncthreads();
    auto key_at = [&](u32 j) -> K { return ((u32)((int)j - w0) <= (u32)(LBVH_BLOCK * 3)) ? s_keys[(int)j - w0] : k[j]; };
    auto delta2p = [&](u32 i, u32 j) -> int {                  // countCommonPrefixBits as used at :52-54 (u32: 32 + clz(i^j) / clz(a^b))
        const K a = key_at(i), b = key_at(j);
        return (a == b) ? ((int)sizeof(K) * 8 + clz_u32(i ^ j)) : (sizeof(K) == 4 ? clz_u32((u32)(a ^ b)) : clz_u64((u64)(a ^ b)));
    };
    const u32 g = blockIdx.x * LBVH_BLOCK + threadIdx.x;
    if (g >= n) return;
    const u32 ni = n - 1;
    {   // InitBvhNodesPrimRef (:164-194): leaf = {primRef.primIdx, INVALID, primRef.aabb}; PrimRef i = {i, bounds(tri i)}
        const u32 prim = svals[g];
        node_store_plain(nodes + ni + g, prim, INV, box_load(boxes + prim));
    }
    if (g >= ni) return;
    const u32 idx = g;
    u32 first, last;
    if (idx == 0) { first = 0; last = n - 1; parent[0] = INV; }
    else {
        const int ld = delta2p(idx, idx - 1), rd = delta2p(idx, idx + 1);
        const int d = (rd > ld) ? 1 : -1;
        const int dmin = (ld < rd) ? ld : rd;
        auto probe = [&](long long jj) -> int { return (jj >= 0 && jj < (long long)n) ? delta2p(idx, (u32)jj) : -1; };
        long long lmax = 2;
        while (probe((long long)idx + d * lmax) > dmin) lmax <<= 1;
        long long l = 0;
        for (long long t = lmax >> 1; t > 0; t >>= 1)
            if (probe((long long)idx + (l + t) * d) > dmin) l += t;
        const u32 jdx = (u32)((long long)idx + l * d);
        if (d < 0) { first = jdx; last = idx; } else { first = idx; last = jdx; }
    }
    const u32 dnode = (u32)delta2p(first, last);
    int split = (int)first, stride = (int)(last - first);
    do {
        stride = (stride + 1) >> 1;
        const int mid = split + stride;
        if ((u32)mid < last && (u32)delta2p(first, (u32)mid) > dnode) split = mid;
    } while (stride > 1);
    const u32 s = (u32)split;
    const u32 lc = (s == first) ? s + ni : s;                // :210-211
    const u32 rc = (s + 1 == last) ? s + 1 + ni : s + 1;
    nodes[idx].left = lc; nodes[idx].right = rc;
    parent[lc] = idx; parent[rc] = idx;
}

__global__ __launch_bounds__(LBVH_BLOCK) void k_refit(bvh2_node* nodes, const u32* __restrict__ parent, u32* flags, u32 n) {
    const u32 g = blockIdx.x * LBVH_BLOCK + threadIdx.x;
    if (g >= n) return;
    const u32 ni = n - 1;
    u32 cur = ni + g;
    Box box = box_load(&nodes[cur].aabb);                   // leaf box: written by k_karras (previous launch)
    u32 p = parent[cur];
    while (p != INV) {
        drain_stores();
        // the reference counts arrivals (atomicAdd(flags) > 0, :224) and then re-reads the child links to find the sibling;
        // exchanging the arriving child's index instead hands the sibling to the second arriver directly
        const u32 sib = __hip_atomic_exchange(flags + p, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sib == INV) break;
        compiler_fence();
        box = box_union(box, node_box_agent(nodes + sib));
        node_box_store_agent(nodes + p, box);
        cur = p; p = parent[p];
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------
// key_bits: 32 = u32 keys (the reference's 30-bit codes), 64 = u64 keys (60-bit codes)
void launch_lbvh_single(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                        void* d_nodes, uint64_t* d_slots, uint32_t* d_root) {
    (void)hipMemsetAsync(d_slots, 0xFF, (size_t)n * sizeof(u64), s);
    const u32 blocks = (n + LBVH_BLOCK - 1) / LBVH_BLOCK;
    KernelScope ks(s, "k_lbvh_single");
    if (key_bits == 64) hipLaunchKernelGGL(k_lbvh_single<u64>, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (const bvh_aabb*)d_boxes, (const u64*)d_skeys, d_svals, (bvh2_node*)d_nodes, d_slots, d_root, n);
    else                hipLaunchKernelGGL(k_lbvh_single<u32>, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (const bvh_aabb*)d_boxes, (const u32*)d_skeys, d_svals, (bvh2_node*)d_nodes, d_slots, d_root, n);
}

void launch_lbvh_two(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                     void* d_nodes, uint32_t* d_parent, uint32_t* d_flags) {
    (void)hipMemsetAsync(d_flags, 0xFF, (size_t)n * sizeof(u32), s);
    const u32 blocks = (n + LBVH_BLOCK - 1) / LBVH_BLOCK;
    { KernelScope ks(s, "k_karras");
      if (key_bits == 64) hipLaunchKernelGGL(k_karras<u64>, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (const bvh_aabb*)d_boxes, (const u64*)d_skeys, d_svals, (bvh2_node*)d_nodes, d_parent, n);
      else                hipLaunchKernelGGL(k_karras<u32>, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (const bvh_aabb*)d_boxes, (const u32*)d_skeys, d_svals, (bvh2_node*)d_nodes, d_parent, n); }
    { KernelScope ks(s, "k_refit"); hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (bvh2_node*)d_nodes, (const u32*)d_parent, d_flags, n); }
}

} // namespace bvh
