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
#include <cstdlib>
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

constexpr int LBVH_BLOCK = 256;
// hand-off words: 0 = empty; {node index + 1, far end of the range}.  The second arriver resets the word, so the array is
// all-zero again after every completed build (it is zeroed once, when the arena is allocated) — no per-build memset.
constexpr u64 SLOT_EMPTY = 0ull;
__device__ __forceinline__ u64 slot_word(u32 node, u32 far) { return ((u64)(node + 1u) << 32) | far; }

// ------------------------------------------------------------------------------------------------------------------
// Single pass.  One walker per leaf.  A finished node covering sorted positions [i,j) chooses its parent exactly as
// findParent (:64-86): parent = j-1 (as left child) if i==0 or (j!=n and delta(j-1,j) < delta(i-1,i)), else i-1 (as right
// child), delta = xor of the 64-bit {key,position} words.  Instead of the reference's {child link store, span store,
// counter atomicAdd, re-load} the two children of a node meet through ONE 64-bit atomic exchange on slots[parent]:
// the first arriver leaves {its node index, the far end of its range} and retires; the second arriver receives it, so it
// knows both children and the parent's full range, reads the sibling's box, and writes the parent node once (32 B).
// ------------------------------------------------------------------------------------------------------------------
constexpr u32 LBQ_SUB = 64;        // sub-queues of the block kernel's hand-over (one atomic per block per launch)
#ifndef LBVH_TILE_SIZE
#define LBVH_TILE_SIZE 512
#endif
constexpr int LBVH_TILE = LBVH_TILE_SIZE;     // leaves per tile of the block schedulers

// Node numbering.  Single pass (Apetrei, src/SinglePassLbvhKernel.h): an internal node's index is its split position.  Two pass (Karras,
// src/TwoPassLbvhKernel.h:196-216): node i covers a range that has i at one end — equivalently (the children of a node with split s are
// `s` and `s+1`, :210-211) a left child's index is the LAST position of its range, a right child's the FIRST, the root is 0.  Both are the
// same tree, so the same bottom-up walk emits either array: a finished node learns its own Karras index when it learns which side of its
// parent it is, one step after it was merged — its record is therefore written at the start of the next step in both numberings.
struct LbvhWalker { u32 i, j, cur, lc, rc; Box box; bool leaf; };      // finished node covers sorted positions [i, j)

// the global part of a walker's climb: second-arriver hand-off on slots[parent], agent-scope (src/SinglePassLbvhKernel.h:88-126)
template <typename K, bool KARRAS>
__device__ __forceinline__ void lbvh_climb(LbvhWalker w, const K* __restrict__ skeys, bvh2_node* nodes, u64* slots, u32* root_out, u32 n) {
    while (true) {
        const bool root = w.i == 0 && w.j == n;
        bool as_left = true;                                // findParent (:64-86)
        if (!root) { if (w.i == 0) as_left = true; else if (w.j == n) as_left = false; else as_left = closer(skeys, w.j - 1, w.i - 1); }
        if (!w.leaf) {
            if (KARRAS) w.cur = root ? 0u : (as_left ? w.j - 1 : w.i);
            node_store_agent(nodes + w.cur, w.lc, w.rc, w.box);
        }
        if (root) { *root_out = w.cur; break; }             // :73 -> root (:116-120)
        const u32 p = as_left ? w.j - 1 : w.i - 1;
        drain_stores();                                     // my node is in memory before anybody can learn its index
        const u64 other = __hip_atomic_exchange(slots + p, slot_word(w.cur, as_left ? w.i : w.j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (other == SLOT_EMPTY) break;                     // first arriver retires (atomicAdd(...) > 0 fails, :103)
        st_agent(slots + p, SLOT_EMPTY);                    // nobody touches this word again: leave it clean for the next build
        compiler_fence();
        const u32 sib = (u32)(other >> 32) - 1u, far = (u32)other;
        w.box = box_union(w.box, node_box_agent(nodes + sib));   // merge(left.aabb, right.aabb) (:112) — min/max commute
        w.lc = as_left ? w.cur : sib; w.rc = as_left ? sib : w.cur; w.leaf = false;
        if (as_left) w.j = far; else w.i = far;
        w.cur = p;                                          // (Apetrei numbering; Karras: decided next step)
    }
}

template <typename K>
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_single(const bvh_aabb* __restrict__ boxes, const K* __restrict__ skeys,
                                                            const u32* __restrict__ svals, bvh2_node* nodes, u64* slots,
                                                            u32* root_out, u32 n) {
    const u32 g = bid_x() * LBVH_BLOCK + tid_x();
    if (g >= n) return;
    const u32 ni = n - 1;
    const u32 prim = svals[g];
    Box box = box_gather(boxes + prim);                       // = bounds of Triangle[prim] (:44), computed once in stage E
    node_store_agent(nodes + ni + g, prim, INV, box);       // leaf record {left = primIdx, right = INVALID} (:36-45)
    lbvh_climb<K, false>(LbvhWalker{ g, g + 1, ni + g, 0u, 0u, box, true }, skeys, nodes, slots, root_out, n);
}

// ------------------------------------------------------------------------------------------------------------------
// Single pass, large inputs: a workgroup owns a tile of T sorted leaves and builds every node whose range lies inside the tile
// with the same second-arriver rule, but through LDS (an LDS exchange per node, the first arriver's box parked in LDS): no global
// atomics, no dependent global loads — the kernel streams {sorted value, box gather} in and {leaf node, internal node} out.  Whether
// the node of gap p leaves the tile is two key probes (does the leaf just outside the tile share p's prefix?).  The subtree roots
// whose parent leaves the tile (the ancestors of the T-aligned gaps: ~4 % of the nodes) are queued; k_lbvh_ext continues their
// climb with the global protocol above.  Node index = split position, as before: the array is byte-identical.
// ------------------------------------------------------------------------------------------------------------------
template <typename K, int T, bool KARRAS>
__global__ __launch_bounds__(T) void k_lbvh_block(const bvh_aabb* __restrict__ boxes, const K* __restrict__ skeys,
                                                  const u32* __restrict__ svals, bvh2_node* __restrict__ nodes,
                                                  uint4* __restrict__ queue, u32* root_out, u32 n) {
    __shared__ K s_key[T + 2];                       // sorted keys of positions g0-1 .. g0+T
    __shared__ u64 s_slot[T];                        // per gap: hand-off word of the first arriver (slot_word), 0 = nobody yet
    __shared__ float s_box[2][6][T];                 // per gap: the box its left / right child parked before the hand-off
    __shared__ unsigned char s_ext[T];               // per gap: the node's range leaves the tile
    __shared__ u64 s_q[T];                           // subtree roots handed to k_lbvh_ext: {node : 32 | i - g0 : 16 | j - g0 : 16}
    __shared__ unsigned short s_inv[KARRAS ? T : 1]; // Karras numbering: node index - g0 -> gap whose node carries it (0xFFFF: none in this tile)
    __shared__ u32 s_nq;
    const int tid = tid_x();
    const u32 ni = n - 1;
    const u32 g0 = bid_x() * (u32)T, g = g0 + (u32)tid;
    const u32 t_end = (g0 + (u32)T < n) ? g0 + (u32)T : n;            // the tile's leaves: [g0, t_end)
    // the key window (T + 2 keys: two loads per thread) and the leaf's primitive index are requested together, the box behind them: two dependent memory round trips.
    // (Indices clamped instead of branched around: the loop over the window plus the branch per leaf came out as four — load, wait, load, wait, index, wait, box; round 4, ISA.)
    constexpr int KW = (T + 2 + T - 1) / T;
    K kw[KW];
#pragma unroll
    for (int q = 0; q < KW; ++q) {
        const int k = tid + q * T;
        const long long j = (long long)g0 - 1 + k;
        const bool in = k < T + 2 && j >= 0 && j < (long long)n;
        kw[q] = skeys[in ? j : (long long)g0];
        if (!in) kw[q] = (K)0;
    }
    const u32 prim = svals[g < n ? g : ni];
#pragma unroll
    for (int q = 0; q < KW; ++q) { const int k = tid + q * T; if (k < T + 2) s_key[k] = kw[q]; }
    s_slot[tid] = 0ull;
    if (KARRAS) s_inv[tid] = 0xFFFFu;
    if (tid == 0) s_nq = 0u;
    Box box = box_empty();
    if (g < n) {
        box = box_gather(boxes + prim);                                     // = bounds of Triangle[prim] (:44), computed once in stage E
        node_store_plain(nodes + ni + g, prim, INV, box);                 // leaf record (:36-45); read again only by later launches
    }
    __syncthreads();
    auto wkey = [&](u32 j) -> K { return s_key[j - g0 + 1u]; };            // j in [g0-1, g0+T]
    {   // does the node of gap g leave the tile?  (the positions sharing its prefix are contiguous: probe the first leaf outside)
        bool ext = true;                                                  // the tile's last gap belongs to two tiles
        if (g + 1u < t_end) {
            const K kp = wkey(g);
            const int c0 = plen(kp, g, wkey(g + 1u), g + 1u);
            ext = (g0 > 0u && plen(wkey(g0 - 1u), g0 - 1u, kp, g) >= c0) || (t_end < n && plen(wkey(t_end), t_end, kp, g) >= c0);
        }
        s_ext[tid] = ext ? 1 : 0;
    }
    __syncthreads();
    if (g < n) {
        u32 i = g, j = g + 1u, cur = ni + g, lc = 0u, rc = 0u;            // finished node `cur` covers sorted positions [i, j)
        bool leaf = true;
        while (true) {
            const bool root = i == 0u && j == n;                          // (single-tile input)
            bool as_left = true;                                          // findParent (:64-86); plen comparison == closer()
            if (!root) {
                if (i == 0u) as_left = true;
                else if (j == n) as_left = false;
                else as_left = plen(wkey(j - 1u), j - 1u, wkey(j), j) > plen(wkey(i - 1u), i - 1u, wkey(i), i);
            }
            // (the finished node's record is not stored here — a few lanes at a time, 32 bytes each, all over the tile's 16 KB: 0.14 ms of a
            //  10 M build — but rebuilt from LDS by the tile's last step: children in s_slot, box = union of the two parked child boxes)
            if (KARRAS && !leaf) { const u32 gap = cur; cur = root ? 0u : (as_left ? j - 1u : i); s_inv[cur - g0] = (unsigned short)(gap - g0); }
            if (root) { *root_out = cur; break; }
            const u32 p = as_left ? j - 1u : i - 1u;
            if (p < g0 || s_ext[p - g0]) {                                // parent leaves the tile: hand the subtree root over
                s_q[atomicAdd(&s_nq, 1u)] = ((u64)cur << 32) | ((u64)(i - g0) << 16) | (u64)(j - g0);
                break;
            }
            const u32 ps = p - g0;
            const int side = as_left ? 0 : 1;                             // park my box, then publish (LDS operations of a wave execute in order)
            s_box[side][0][ps] = box.lx; s_box[side][1][ps] = box.ly; s_box[side][2][ps] = box.lz; s_box[side][3][ps] = box.hx; s_box[side][4][ps] = box.hy; s_box[side][5][ps] = box.hz;
            compiler_fence();                                             // the compiler may not sink the box stores below the exchange ...
            const u64 other = atomicExch(reinterpret_cast<unsigned long long*>(&s_slot[ps]), (unsigned long long)slot_word(cur, as_left ? i : j));
            compiler_fence();                                             // ... nor hoist the sibling's box loads above it
            if (other == SLOT_EMPTY) break;                               // first arriver retires
            const u32 sib = (u32)(other >> 32) - 1u, far = (u32)other;
            const Box sb = { s_box[1 - side][0][ps], s_box[1 - side][1][ps], s_box[1 - side][2][ps], s_box[1 - side][3][ps], s_box[1 - side][4][ps], s_box[1 - side][5][ps] };
            box = box_union(box, sb);                                     // merge(left.aabb, right.aabb) (:112) — min/max commute
            lc = as_left ? cur : sib; rc = as_left ? sib : cur; leaf = false;
            s_slot[ps] = (u64)lc | ((u64)rc << 32);                       // (the word has seen both arrivals: it now keeps the children)
            if (as_left) j = far; else i = far;
            cur = p;                                                      // (Apetrei numbering; Karras: decided next step)
        }
    }
    __syncthreads();
    {   // the tile's internal records, one per thread in index order: full 32-byte-per-lane rows instead of the climb's scattered stores
        const u32 gap = KARRAS ? (u32)s_inv[tid] : (u32)tid;
        if (KARRAS ? gap != 0xFFFFu : !s_ext[tid]) {
            const Box l = { s_box[0][0][gap], s_box[0][1][gap], s_box[0][2][gap], s_box[0][3][gap], s_box[0][4][gap], s_box[0][5][gap] };
            const Box r = { s_box[1][0][gap], s_box[1][1][gap], s_box[1][2][gap], s_box[1][3][gap], s_box[1][4][gap], s_box[1][5][gap] };
            const u64 ch = s_slot[gap];
            node_store_plain(nodes + g0 + (u32)tid, (u32)ch, (u32)(ch >> 32), box_union(l, r));
        }
    }
    // the tile's roots go to the tile's own segment of the queue (T entries; entry 0 carries the count in .w): no atomic, no round trip at the
    // end of the workgroup (a returning atomic on a sub-queue head here was ~2 us of a ~20 us workgroup)
    const u32 nq = s_nq;
    uint4* seg = queue + (size_t)bid_x() * (size_t)T;
    if (tid == 0 && nq == 0u) seg[0] = make_uint4(0u, 0u, 0u, 0u);
    for (u32 k = (u32)tid; k < nq; k += (u32)T) {
        const u64 it = s_q[k];
        seg[k] = make_uint4((u32)(it >> 32), g0 + (u32)((it >> 16) & 0xFFFFu), g0 + (u32)(it & 0xFFFFu), k == 0u ? nq : 0u);
    }
}

// the subtree roots queued by k_lbvh_block (records written, boxes in memory) continue with the global second-arriver protocol
template <typename K, bool KARRAS>
__global__ __launch_bounds__(LBVH_BLOCK) void k_lbvh_ext(const K* __restrict__ skeys, bvh2_node* nodes, u64* slots, const uint4* __restrict__ queue,
                                                         u32 slot_shift, u32* root_out, u32 n) {
    // 2^slot_shift lanes per tile segment (a tile hands over 2-3 roots on average, at most LBVH_TILE).  The climb is a chain of coherent round trips
    // per lane and a wave runs as long as its longest lane: FEW roots per wave and many waves is what is fast — 64 roots per wave from a dense
    // queue took 0.123 ms at 10 M; 8 / 16 / 32 lanes per tile (about 20 / 10 / 5 roots per wave): 0.088 / 0.076 / 0.084 ms (32: more threads than
    // the chip holds), at 2 M 0.062 / 0.048 / 0.042.
    const u32 ntiles = (n + (u32)LBVH_TILE - 1u) / (u32)LBVH_TILE;
    const u32 lanes = 1u << slot_shift;
    for (u32 idx = bid_x() * LBVH_BLOCK + tid_x(); idx < (ntiles << slot_shift); idx += nbid_x() * LBVH_BLOCK) {
        const uint4* seg = queue + (size_t)(idx >> slot_shift) * (size_t)LBVH_TILE;
        const u32 cnt = seg[0].w;
        for (u32 k = idx & (lanes - 1u); k < cnt; k += lanes) {
            const uint4 it = seg[k];
            const Box box = box_load(&nodes[it.x].aabb);                  // written by k_lbvh_block (previous launch)
            // (handed over as "leaf": its record exists already, only its parent's side is still to be found)
            lbvh_climb<K, KARRAS>(LbvhWalker{ it.y, it.z, it.x, 0u, 0u, box, true }, skeys, nodes, slots, root_out, n);
        }
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
    const int g0 = (int)(bid_x() * LBVH_BLOCK), w0 = g0 - LBVH_BLOCK;
    for (int t = tid_x(); t < LBVH_BLOCK * 3 + 1; t += LBVH_BLOCK) { const int j = w0 + t; s_keys[t] = (j >= 0 && j < (int)n) ? k[j] : (K)0; }
    __syncthreads();
    auto key_at = [&](u32 j) -> K { return ((u32)((int)j - w0) <= (u32)(LBVH_BLOCK * 3)) ? s_keys[(int)j - w0] : k[j]; };
    auto delta2p = [&](u32 i, u32 j) -> int {                  // countCommonPrefixBits as used at :52-54 (u32: 32 + clz(i^j) / clz(a^b))
        const K a = key_at(i), b = key_at(j);
        return (a == b) ? ((int)sizeof(K) * 8 + clz_u32(i ^ j)) : (sizeof(K) == 4 ? clz_u32((u32)(a ^ b)) : clz_u64((u64)(a ^ b)));
    };
    const u32 g = bid_x() * LBVH_BLOCK + tid_x();
    if (g >= n) return;
    const u32 ni = n - 1;
    {   // InitBvhNodesPrimRef (:164-194): leaf = {primRef.primIdx, INVALID, primRef.aabb}; PrimRef i = {i, bounds(tri i)}
        const u32 prim = svals[g];
        node_store_plain(nodes + ni + g, prim, INV, box_gather(boxes + prim));
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

// the refit walk of one finished node `cur` (its box is in memory) through the global second-arriver protocol.  The reference counts
// arrivals (atomicAdd(flags) > 0, :224) and then re-reads the child links to find the sibling; exchanging the arriving child's index
// instead hands the sibling to the second arriver directly.  The second arriver resets the word: the array stays all-INVALID.
__device__ __forceinline__ void refit_climb(u32 cur, Box box, bvh2_node* nodes, const u32* __restrict__ parent, u32* flags) {
    u32 p = parent[cur];
    while (p != INV) {
        drain_stores();
        const u32 sib = __hip_atomic_exchange(flags + p, cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sib == INV) break;
        st_agent(flags + p, INV);
        compiler_fence();
        box = box_union(box, node_box_agent(nodes + sib));
        node_box_store_agent(nodes + p, box);
        cur = p; p = parent[p];
    }
}

__global__ __launch_bounds__(LBVH_BLOCK) void k_refit(bvh2_node* nodes, const u32* __restrict__ parent, u32* flags, u32 n) {
    const u32 g = bid_x() * LBVH_BLOCK + tid_x();
    if (g >= n) return;
    const u32 cur = n - 1 + g;
    refit_climb(cur, box_load(&nodes[cur].aabb), nodes, parent, flags);    // leaf box: written by k_karras (previous launch)
}

// ---- launchers ---------------------------------------------------------------------------------------------------
// key_bits: 32 = u32 keys (the reference's 30-bit codes), 64 = u64 keys (60-bit codes).  d_slots: u64[n], all-zero (kept clean by the
// protocol).  Large inputs: tile kernel + external climb; d_queue: uint4[queue_capacity], d_queue_count: u32[64 * 32 + 1].
constexpr uint32_t LBVH_BLOCK_MIN_N = 240000;      // below: one launch (k_lbvh_single / k_refit).  Whole single-pass build, one launch / tiles (tools/ab_sched_small.py, end of round 3):
                                                   // 150 k 0.1022 / 0.1074 ms, 200 k 0.1081 / 0.1104, Sponza-like 262 144 0.1175 / 0.1147, uniform 262 144 0.1136 / 0.1112, 400 k 0.1339 / 0.1285
size_t lbvh_queue_capacity(uint32_t n) { return (((size_t)n / LBVH_TILE + 1) / LBQ_SUB + 2) * LBVH_TILE * LBQ_SUB; }   // every tile may queue T roots
static bool lbvh_use_tiles(uint32_t n, int scheduler) {   // scheduler: BVH_OPT_LBVH_SCHEDULER (1 one-launch kernels, 2 tiles: the host's override)
    return scheduler == 2 ? true : scheduler == 1 ? false : n >= LBVH_BLOCK_MIN_N;
}
// tile kernel + external climb; karras: emit the two-pass builder's node numbering instead of the single-pass one
static void launch_lbvh_tiles(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n, void* d_nodes,
                              uint64_t* d_slots, uint32_t* d_root, void* d_queue, size_t queue_capacity, uint32_t* d_queue_count, bool heads_cleared, bool karras) {
    (void)queue_capacity; (void)d_queue_count; (void)heads_cleared;   // (every tile owns a segment of the queue: no heads to clear)
    const u32 ntiles = (n + LBVH_TILE - 1) / LBVH_TILE;
#ifndef LBVH_EXT_MAX_SHIFT
#define LBVH_EXT_MAX_SHIFT 5
#endif
    u32 slot_shift = LBVH_EXT_MAX_SHIFT;                // lanes per tile segment in k_lbvh_ext: as many as keep every thread resident (<= 400 k threads), 8..32
    while (slot_shift > 3 && ((size_t)ntiles << slot_shift) > 400000u) --slot_shift;
    const u32 eb = ((ntiles << slot_shift) + LBVH_BLOCK - 1) / LBVH_BLOCK;
    const dim3 gt(ntiles), bt(LBVH_TILE), ge(eb < 4096u ? eb : 4096u), be(LBVH_BLOCK);
#define LB_TILES(KK, KAR) do { \
        { KernelScope ks(s, "k_lbvh_block"); hipLaunchKernelGGL((k_lbvh_block<KK, LBVH_TILE, KAR>), gt, bt, 0, s, (const bvh_aabb*)d_boxes, (const KK*)d_skeys, d_svals, \
                                                                (bvh2_node*)d_nodes, (uint4*)d_queue, d_root, n); } \
        { KernelScope ks(s, "k_lbvh_ext"); hipLaunchKernelGGL((k_lbvh_ext<KK, KAR>), ge, be, 0, s, (const KK*)d_skeys, (bvh2_node*)d_nodes, d_slots, (const uint4*)d_queue, \
                                                              slot_shift, d_root, n); } } while (0)
    if (key_bits == 64) { if (karras) LB_TILES(u64, true); else LB_TILES(u64, false); }
    else                { if (karras) LB_TILES(u32, true); else LB_TILES(u32, false); }
#undef LB_TILES
}

void launch_lbvh_single(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                        void* d_nodes, uint64_t* d_slots, uint32_t* d_root, void* d_queue, size_t queue_capacity, uint32_t* d_queue_count, bool heads_cleared, int scheduler) {
    if (!lbvh_use_tiles(n, scheduler) || !d_queue || queue_capacity < lbvh_queue_capacity(n)) {   // (a tile may queue up to T roots: never run the tile kernel on a smaller queue)
        const u32 blocks = (n + LBVH_BLOCK - 1) / LBVH_BLOCK;
        KernelScope ks(s, "k_lbvh_single");
        if (key_bits == 64) hipLaunchKernelGGL(k_lbvh_single<u64>, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (const bvh_aabb*)d_boxes, (const u64*)d_skeys, d_svals, (bvh2_node*)d_nodes, d_slots, d_root, n);
        else                hipLaunchKernelGGL(k_lbvh_single<u32>, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (const bvh_aabb*)d_boxes, (const u32*)d_skeys, d_svals, (bvh2_node*)d_nodes, d_slots, d_root, n);
        return;
    }
    launch_lbvh_tiles(s, d_boxes, d_skeys, key_bits, d_svals, n, d_nodes, d_slots, d_root, d_queue, queue_capacity, d_queue_count, heads_cleared, false);
}

// Two pass.  Small inputs: k_karras + k_refit (d_parent: u32[2n-1]; d_flags: u32[n] exchange words, all-INVALID before the call and left so).
// Large inputs: the same tree is the single-pass tree with another numbering, so the tile scheduler emits it directly (no range
// searches, no parent array): d_slots / d_root / d_queue / d_queue_count as for launch_lbvh_single.
void launch_lbvh_two(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                     void* d_nodes, uint32_t* d_parent, uint32_t* d_flags, uint64_t* d_slots, uint32_t* d_root, void* d_queue, size_t queue_capacity,
                     uint32_t* d_queue_count, bool heads_cleared, int scheduler) {
    if (lbvh_use_tiles(n, scheduler) && d_queue && queue_capacity >= lbvh_queue_capacity(n)) {
        launch_lbvh_tiles(s, d_boxes, d_skeys, key_bits, d_svals, n, d_nodes, d_slots, d_root, d_queue, queue_capacity, d_queue_count, heads_cleared, true);
        return;
    }
    const u32 blocks = (n + LBVH_BLOCK - 1) / LBVH_BLOCK;
    { KernelScope ks(s, "k_karras");
      if (key_bits == 64) hipLaunchKernelGGL(k_karras<u64>, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (const bvh_aabb*)d_boxes, (const u64*)d_skeys, d_svals, (bvh2_node*)d_nodes, d_parent, n);
      else                hipLaunchKernelGGL(k_karras<u32>, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (const bvh_aabb*)d_boxes, (const u32*)d_skeys, d_svals, (bvh2_node*)d_nodes, d_parent, n); }
    { KernelScope ks(s, "k_refit"); hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(LBVH_BLOCK), 0, s, (bvh2_node*)d_nodes, (const u32*)d_parent, d_flags, n); }
}

// (kernels.hpp: touching one kernel of this translation unit makes the runtime load its code object — bvh_ctx_create does that for the build path's modules, so
// that a context's FIRST build does not pay for it: 0.3-0.7 ms per module on the MI355X, tools/cold_probe.py)
void warm_lbvh() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_refit)); }

} // namespace bvh
