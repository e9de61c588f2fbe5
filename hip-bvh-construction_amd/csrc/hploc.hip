// hploc.hip — stage B for HPLOC on gfx950 (wave64).
//
// Replaces SetupClusters + HPloc of the reference (src/HplocKernel.h:39-56, :66-81 findParent, :83-117
// findNearestNeighbours, :126-190 mergeClusters, :192-218 load/storeIndices, :220-255 plocMerge, :257-315 HPloc; host
// src/Hploc.cpp:83-121).  Output: Bvh2Node[n-1] (root = node 0) + PrimRef[n] leaves in Morton order.  The tree topology is
// the reference's: the implicit LBVH hierarchy over the sorted {key,position} words decides WHICH ranges are merged, and
// every range larger than 16 leaves (or the root) runs PLOC rounds on the <= 16 + 16 surviving clusters of its two children
// (search radius 8, mutual nearest neighbours under the {area bits, slot} order, lower slot owns the merge) until <= 16
// (root: 1) remain.
//
// How the work is organised here (MI355X-first, not the reference's per-leaf walker):
//  * one thread per LBVH gap p (between sorted leaves p and p+1) computes that node's leaf range [L,R] straight from the
//    keys (common-prefix search; Karras/Apetrei trees are the same tree).  Ranges of <= 16 leaves need no work at all —
//    the reference walks them with two global atomics per node just to discover them.
//  * only "big" nodes (range > 16, ~n/11 of them) take part in a dependency protocol: counter[p] reaches 3 when the node's
//    own thread (contributing 3 - #big children) and each big child (contributing 1) have arrived; whoever completes the
//    count runs the merge task.  ~2 agent-scope atomics per big node instead of 2 per node.
//  * a merge task occupies one 32-lane half of a wave64: the work list (id, rep, box) lives in registers, neighbours come
//    from DPP wave shifts, partners through ds_bpermute, compaction through ds_permute.  Two tasks run side by side in the
//    two halves.  No LDS allocation, no barriers, no reliance on store conflict order.
//  * SetupClusters is fused: a leaf's PrimRef record is written when the leaf is first loaded as a cluster (exactly once);
//    untouched child ranges are implicit (cluster id = n-1 + position), so the cluster-id array needs no initialisation.
//  * node allocation: the reference takes node indices from ONE global counter (:163-167); a single word sustains ~90
//    returning atomics/us on MI355X (~23 ms for a 10 M build).  Here every cluster carries the sorted position of its first
//    leaf ("rep"); lists stay ordered by rep, a merge keeps the lower partner's rep and retires the absorbed partner's rep
//    r in [1,n) exactly once — node index r-1 is a bijection onto [0,n-1).  The final merge moves whatever occupies node 0 to
//    its own natural slot so that the root is node 0, as the reference guarantees.  Numbering depends on the topology only.
//
// Hand-off between waves (possibly on different XCDs): survivors (cidx) and node boxes written during the launch are
// agent-scope write-through stores read back with agent-scope loads; a wave drains its stores (s_waitcnt vmcnt(0)) before
// the agent-scope atomic on counter[] that publishes a finished range.
#include <cstdlib>
#include <cstdio>
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

// ablation switch for measurements only (results are wrong when set): compiled in with -DBVH_ABLATION
static inline int hploc_ablation() {
    const char* e = getenv("BVH_HPLOC_DEBUG"); return e ? atoi(e) : 0;
}

constexpr int HP_BLOCK = 256;
constexpr u32 HP_HALF = 16;        // WarpSize/2 of the reference's wave32 (src/HplocKernel.h:195,238)
constexpr int HP_RADIUS = 8;       // PlocRadius, src/Common.h:595

__device__ __forceinline__ int clz64(u64 v) { return v ? __clzll((long long)v) : 64; }
__device__ __forceinline__ float dpp_shl1(float v) {   // lane i <- lane i+1 (whole wave; DPP wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xF, 0xF, true));   // bound_ctrl: lane 63 reads 0
}
__device__ __forceinline__ Box box_shl1(const Box& b) { return { dpp_shl1(b.lx), dpp_shl1(b.ly), dpp_shl1(b.lz), dpp_shl1(b.hx), dpp_shl1(b.hy), dpp_shl1(b.hz) }; }
__device__ __forceinline__ Box shfl_box(const Box& b, int src) {
    return { __shfl(b.lx, src), __shfl(b.ly, src), __shfl(b.lz, src), __shfl(b.hx, src), __shfl(b.hy, src), __shfl(b.hz, src) };
}
// push semantics: lane l's value lands in lane dst(l)
__device__ __forceinline__ u32 push_u32(int dst, u32 v) { return (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)v); }
__device__ __forceinline__ float push_f32(int dst, float v) { return __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(v))); }

// cidx entry: {cluster id, rep}
__device__ __forceinline__ u64 entry(u32 id, u32 rep) { return (u64)id | ((u64)rep << 32); }

// ---- one merge pass: every 32-lane half with `have` runs the plocMerge (:220-255) of the LBVH node at gap tP with leaf
// range [tL, tR].  AGENT = true: survivors / node boxes may have been written by other workgroups of the same launch
// (agent-scope loads and write-through stores); AGENT = false: everything read was written by earlier launches.
template <bool AGENT> __device__ __forceinline__ u64 ld_e(const u64* p) { return AGENT ? ld_agent(p) : *p; }
template <bool AGENT> __device__ __forceinline__ u32 ld_w(const u32* p) { return AGENT ? ld_agent(p) : *p; }
template <bool AGENT> __device__ __forceinline__ void st_e(u64* p, u64 v) { if (AGENT) st_agent(p, v); else *p = v; }
template <bool AGENT> __device__ __forceinline__ void st_w(u32* p, u32 v) { if (AGENT) st_agent(p, v); else *p = v; }
template <bool AGENT> __device__ __forceinline__ Box node_box(const bvh2_node* n) {
    if (AGENT) return node_box_agent(n);
    const float* f = reinterpret_cast<const float*>(n) + 2;
    return { f[0], f[1], f[2], f[3], f[4], f[5] };
}
template <bool AGENT> __device__ __forceinline__ void node_store(bvh2_node* n, u32 l, u32 r, const Box& b) {
    if (AGENT) { node_store_agent(n, l, r, b); return; }
    node_store_plain(n, l, r, b);
}

struct HpEntry { u32 id, rep, prim; };   // prim: primitive index prefetched for implicit leaves, INV otherwise

// loadIndices (:192-206): slots 0..15 <- left child, 16..31 <- right child; small children are implicit leaves
template <bool AGENT>
__device__ __forceinline__ HpEntry load_entry(bool have, u32 tL, u32 tR, u32 tP, const u32* __restrict__ svals, const u64* cidx, u32 ni, int slot) {
        const bool is_left = slot < 16;
        const u32 s = (u32)(slot & 15);
        const u32 c_start = is_left ? tL : tP + 1, c_len = is_left ? (tP - tL + 1) : (tR - tP);
        HpEntry en = { INV, INV, INV };
        if (have) {
            if (c_len > HP_HALF) { const u64 e = ld_e<AGENT>(cidx + c_start + s); en.id = (u32)e; en.rep = (u32)(e >> 32); }
            else if (s < c_len) { en.rep = c_start + s; en.id = ni + en.rep; en.prim = svals[en.rep]; }
        }
        return en;
}

struct HpWork { u32 id, rep, cnt, tL; Box b; bool have, final_; };

// left-pack the loaded entries, fetch their boxes (leaves: fused SetupClusters), return the work list of this half
template <bool AGENT, bool SETUP = true>
__device__ __forceinline__ HpWork prepare(bool have, u32 tL, u32 tR, HpEntry en, const bvh_aabb* __restrict__ boxes,
                                          const u32* __restrict__ svals, bvh_primref* __restrict__ leaves, const bvh2_node* nodes,
                                          u32 ni, int slot, int hbase) {
        u32 id = en.id, rep = en.rep, prim = en.prim;
        const u32 vb = (u32)(__ballot(id != INV) >> hbase);
        const u32 nl = (u32)__popc(vb & 0xFFFFu), nr = (u32)__popc(vb & 0xFFFF0000u);
        const u32 cnt = nl + nr;
        {   // left-pack: slot t < nl <- slot t ; slot t in [nl, cnt) <- slot 16 + (t - nl)
            const int src = hbase + (((u32)slot < nl) ? slot : (int)((16 + (u32)slot - nl) & 31));
            const u32 ti = (u32)__shfl((int)id, src), tr = (u32)__shfl((int)rep, src), tp = (u32)__shfl((int)prim, src);
            id = ((u32)slot < cnt) ? ti : INV; rep = tr; prim = tp;
        }
        Box b = box_empty();
        if (id != INV) {
            if (id >= ni) {   // first (and only) load of this leaf: fused SetupClusters (:44-47)
                if (prim == INV) prim = svals[rep];
                b = box_load(boxes + prim);
                if (SETUP) {  // (SETUP = false: the block-local kernel wrote every PrimRef while staging its leaves)
                    float* f = reinterpret_cast<float*>(leaves + rep);
                    reinterpret_cast<u32*>(f)[0] = prim;
                    f[1] = b.lx; f[2] = b.ly; f[3] = b.lz; f[4] = b.hx; f[5] = b.hy; f[6] = b.hz;
                }
            } else b = node_box<AGENT>(nodes + id);                                                  // :242-246
        }
        HpWork w; w.id = id; w.rep = rep; w.cnt = cnt; w.tL = tL; w.b = b; w.have = have; w.final_ = have && tL == 0 && tR == ni;
        return w;
}

// PLOC rounds until <= 16 clusters (root: 1) remain; the work list stays in registers (w is updated in place)
template <bool AGENT>
__device__ __forceinline__ void ploc_rounds(HpWork& w, bvh2_node* nodes, u32* zero_parent, u32 ni, int lane, int slot, int hbase, int dbg) {
        const bool have = w.have, final_ = w.final_;
        u32 id = w.id, rep = w.rep, cnt = w.cnt;
        Box b = w.b;
        const u32 threshold = dbg == 2 ? 64u : (final_ ? 1u : HP_HALF);
        while (__ballot(have && cnt > threshold)) {
            const bool act = have && cnt > threshold;
            // findNearestNeighbours (:83-117): minimum of {area bits, neighbour slot}; each pair's area is evaluated once.
            // Two running minima instead of one 64-bit key: right candidates arrive with increasing slot (strict < keeps the
            // lower slot on ties), left candidates with decreasing slot (<= takes the lower slot), left beats right on ties.
            u32 abR = 0xFFFFFFFFu, abL = 0xFFFFFFFFu; int idR = 0, idL = 0;
            Box nb = b;
#pragma unroll
            for (int r = 1; r <= HP_RADIUS; ++r) {
                nb = box_shl1(nb);                                               // box of slot + r
                const u32 ab = __float_as_uint(box_area(box_union(nb, b)));
                const u32 ab_left = (u32)__shfl_up((int)ab, r);                  // area(slot - r, slot)
                if ((u32)(slot + r) < cnt && ab < abR) { abR = ab; idR = slot + r; }
                if (slot >= r && (u32)slot < cnt && ab_left <= abL) { abL = ab_left; idL = slot - r; }
            }
            // mergeClusters (:126-190)
            const int nbr = (abL <= abR) ? idL : idR;
            const int nsrc = hbase + nbr;
            const u32 nbr_of_nbr = (u32)__shfl(nbr, nsrc);
            const bool in = act && (u32)slot < cnt;
            const bool mutual = in && nbr_of_nbr == (u32)slot;
            const bool merge = mutual && slot < nbr;
            const bool absorbed = mutual && slot > nbr;
            const u32 id_nb = (u32)__shfl((int)id, nsrc);
            const u32 rep_nb = (u32)__shfl((int)rep, nsrc);
            const Box bn = shfl_box(b, nsrc);
            if (merge) {
                b = box_union(b, bn);
                u32 at = rep_nb - 1u;                            // the absorbed partner's rep is retired here, once
                u32 l = id, r = id_nb;
                if (final_ && cnt == 2u && at != 0u) {
                    // the root must be node 0 (:165-167 makes the last allocation 0): move node 0's occupant to the root's
                    // natural slot and re-point its parent
                    const u64* q0 = reinterpret_cast<const u64*>(nodes);
                    u64* qs = reinterpret_cast<u64*>(nodes + at);
                    const u64 w0 = ld_e<AGENT>(q0 + 0), w1 = ld_e<AGENT>(q0 + 1), w2 = ld_e<AGENT>(q0 + 2), w3 = ld_e<AGENT>(q0 + 3);
                    st_e<AGENT>(qs + 0, w0); st_e<AGENT>(qs + 1, w1); st_e<AGENT>(qs + 2, w2); st_e<AGENT>(qs + 3, w3);
                    if (l == 0u) l = at;
                    else if (r == 0u) r = at;
                    else {
                        const u32 pw = ld_w<AGENT>(zero_parent);
                        st_w<AGENT>(reinterpret_cast<u32*>(nodes + (pw >> 1)) + (pw & 1u), at);
                    }
                    at = 0u;
                } else if (l == 0u || r == 0u) st_w<AGENT>(zero_parent, (at << 1) | (r == 0u ? 1u : 0u));   // who points at node 0
                node_store<AGENT>(nodes + at, l, r, b);
                id = at;
            }
            // compaction: survivors and merged clusters keep their order (:176-187 as "valid slots write to their rank").
            // Slot 31 of a half is never a destination after a round (>= 1 merge), so it serves as the sink.
            const bool keep = in && !absorbed;
            const u32 kh = (u32)(__ballot(keep) >> hbase);
            const u32 newcnt = (u32)__popc(kh);
            const int dst = act ? (hbase + (keep ? (int)__popc(kh & ((1u << slot) - 1u)) : 31)) : lane;
            id = push_u32(dst, id); rep = push_u32(dst, rep);
            b.lx = push_f32(dst, b.lx); b.ly = push_f32(dst, b.ly); b.lz = push_f32(dst, b.lz);
            b.hx = push_f32(dst, b.hx); b.hy = push_f32(dst, b.hy); b.hz = push_f32(dst, b.hz);
            if (act) { if ((u32)slot >= newcnt) id = INV; cnt = newcnt; }
        }
        w.id = id; w.rep = rep; w.cnt = cnt; w.b = b;
}

// PLOC rounds, then storeIndices (:208-218): the <= 16 survivors of a non-root range, INVALID-terminated
template <bool AGENT>
__device__ __forceinline__ void reduce_and_store(HpWork w, bvh2_node* nodes, u64* cidx, u32* zero_parent, u32 ni, int lane, int slot, int hbase, int dbg) {
        ploc_rounds<AGENT>(w, nodes, zero_parent, ni, lane, slot, hbase, dbg);
        if (w.have && !w.final_ && slot < 16) st_e<AGENT>(cidx + w.tL + slot, entry(w.id, w.rep));
}

template <bool AGENT, bool SETUP = true>
__device__ __forceinline__ void merge_exec(bool have, u32 tL, u32 tR, HpEntry en, const bvh_aabb* __restrict__ boxes,
                                           const u32* __restrict__ svals, bvh_primref* __restrict__ leaves, bvh2_node* nodes,
                                           u64* cidx, u32* zero_parent, u32 ni, int lane, int slot, int hbase, int dbg) {
    reduce_and_store<AGENT>(prepare<AGENT, SETUP>(have, tL, tR, en, boxes, svals, leaves, nodes, ni, slot, hbase), nodes, cidx, zero_parent, ni, lane, slot, hbase, dbg);
}

// ---- the asynchronous part: run ready merge tasks, two per pass (one per 32-lane half of the wave), then hand the finished
// range to the parent node; whoever completes the parent's dependency count (3) runs it next.  No waiting anywhere.
// PROP = true: a node's range is assembled bottom-up — the finishing left child stores its L into the low half of ranges[q], the
// right child its R into the high half (the parent's own thread supplies the half of a small child), so nobody searches.
template <bool SETUP, bool PROP = false>
__device__ __forceinline__ void async_climb(bool ready, u32 pc, u32 L, u32 R, const bvh_aabb* __restrict__ boxes, const u32* __restrict__ skeys,
                                            const u32* __restrict__ svals, bvh_primref* __restrict__ leaves, bvh2_node* nodes, u64* cidx,
                                            u64* ranges, u32* counter, u32* zero_parent, u32 ni, int lane, int dbg) {
    const int half = lane >> 5, slot = lane & 31, hbase = half << 5;
    while (true) {
        const u64 rm = __ballot(ready);
        if (!rm) break;
        const int ownA = __ffsll((unsigned long long)rm) - 1;
        const u64 rm2 = rm & (rm - 1);
        const int ownB = rm2 ? __ffsll((unsigned long long)rm2) - 1 : -1;
        const int own = half ? ownB : ownA;
        const bool have = own >= 0;
        const int osrc = have ? own : 0;
        const u32 tL = (u32)__shfl((int)L, osrc), tR = (u32)__shfl((int)R, osrc), tP = (u32)__shfl((int)pc, osrc);
        merge_exec<true, SETUP>(have, tL, tR, load_entry<true>(have, tL, tR, tP, svals, cidx, ni, slot), boxes, svals, leaves, nodes, cidx, zero_parent, ni, lane, slot, hbase, dbg);

        // -- the owners hand their finished range to the parent node
        if (ready && (lane == ownA || lane == ownB)) {
            ready = false;
            if (!(L == 0 && R == ni)) {
                // findParent (:66-81): the boundary gap with the longer common prefix (smaller xor) is the parent
                u32 q;
                if (L == 0) q = R;
                else if (R == ni) q = L - 1;
                else q = ((aug_key(skeys, R) ^ aug_key(skeys, R + 1)) < (aug_key(skeys, L - 1) ^ aug_key(skeys, L))) ? R : L - 1;
                if (PROP) st_agent(reinterpret_cast<u32*>(ranges + q) + (q == R ? 0 : 1), q == R ? L : R);
                drain_stores();                 // the wave's node / survivor stores are in memory before the count moves
                const u32 old = __hip_atomic_fetch_add(counter + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1u == 3u) {
                    compiler_fence();
                    const u64 rg = ld_agent(ranges + q);
                    pc = q; L = (u32)rg; R = (u32)(rg >> 32); ready = true;
                }
            }
        }
    }
}

#ifndef HP_WAVES
#define HP_WAVES 1
#endif
__global__ __launch_bounds__(HP_BLOCK, HP_WAVES) void k_hploc(const bvh_aabb* __restrict__ boxes, const u32* __restrict__ skeys,
                                                    const u32* __restrict__ svals, bvh_primref* __restrict__ leaves,
                                                    bvh2_node* nodes, u64* cidx, u64* ranges, u32* counter, u32* zero_parent, u32 n, int dbg) {
    const int lane = threadIdx.x & (WAVE - 1);
    const u32 ni = n - 1;
    u32 pc = blockIdx.x * HP_BLOCK + threadIdx.x;      // LBVH gap / node this lane currently speaks for
    u32 L = 0, R = 0;
    bool ready = false;

    // ---- phase 1: range of node pc from the keys; dependency bookkeeping for big nodes ---------------------------------
    // The block's key window [g0 - 256, g0 + 512] sits in LDS: almost every probe of the common-prefix searches lands there
    // (a dependent L2 round trip per probe otherwise); only ranges reaching beyond the window probe global memory.
    __shared__ u32 s_keys[HP_BLOCK * 3 + 1];
    const int g0 = (int)(blockIdx.x * HP_BLOCK);
    const int w0 = g0 - HP_BLOCK;
    for (int k = threadIdx.x; k < HP_BLOCK * 3 + 1; k += HP_BLOCK) { const int j = w0 + k; s_keys[k] = (j >= 0 && j < (int)n) ? skeys[j] : 0u; }
    __syncthreads();
    auto key_at = [&](int j) -> u64 {
        const u32 k = ((u32)(j - w0) <= (u32)(HP_BLOCK * 3)) ? s_keys[j - w0] : skeys[j];
        return ((u64)k << 32) | (u32)j;
    };
    if (pc < ni) {
        const int p = (int)pc;
        const u64 kp = key_at(p);
        const int c0 = clz64(kp ^ key_at(p + 1));                               // common prefix length of the node
        auto inside = [&](int j) -> bool { return j >= 0 && j < (int)n && clz64(key_at(j) ^ kp) >= c0; };
        {   // leftmost leaf sharing the prefix (n < 2^30: int arithmetic cannot overflow)
            int step = 1;
            while (inside(p - step)) step <<= 1;
            int lo = p - (step >> 1);                                           // known inside (step 1 -> p itself)
            for (int t = step >> 2; t > 0; t >>= 1) if (inside(lo - t)) lo -= t;
            L = (u32)lo;
        }
        {   // rightmost
            int step = 1;
            while (inside(p + 1 + step)) step <<= 1;
            int hi = p + 1 + (step >> 1);
            for (int t = step >> 2; t > 0; t >>= 1) if (inside(hi + t)) hi += t;
            R = (u32)hi;
        }
        const u32 size = R - L + 1;
        if (size > HP_HALF || size == n) {                                       // :303-305 (size > 16 or root)
            const u32 e = ((pc - L + 1) > HP_HALF ? 1u : 0u) + ((R - pc) > HP_HALF ? 1u : 0u);
            if (e == 0) ready = true;
            else {
                st_agent(ranges + pc, (u64)L | ((u64)R << 32));
                drain_stores();
                const u32 old = __hip_atomic_fetch_add(counter + pc, 3u - e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ready = (old + (3u - e)) == 3u;
            }
        }
    }

    if (dbg == 1) return;
    async_climb<true>(ready, pc, L, R, boxes, skeys, svals, leaves, nodes, cidx, ranges, counter, zero_parent, ni, lane, dbg);
}

// =====================================================================================================================
// Level-synchronous variant for large inputs.
//
// In the LBVH hierarchy a parent's common prefix is strictly shorter than its children's, so "process nodes by decreasing
// prefix length c0" is a topological order that needs no dependency tracking at all: k_hp_plan computes every node's range
// and c0 from the keys, the one-sweep sort (one 8-bit pass) groups the big nodes by level 63 - c0, and one launch per level
// runs that level's merge tasks — two per wave64, always paired, with plain (cached) loads and stores, no atomics, no
// drains: the kernel boundary is the hand-off.  Empty levels cost one ~2 us launch each.
// =====================================================================================================================
constexpr u32 HP_NO_TASK = 0xFFu;

__global__ __launch_bounds__(HP_BLOCK) void k_hp_plan(const u32* __restrict__ skeys, u64* __restrict__ ranges,
                                                      u32* __restrict__ level_keys, u32 n) {
    __shared__ u32 s_keys[HP_BLOCK * 3 + 1];
    const u32 ni = n - 1;
    const u32 pc = blockIdx.x * HP_BLOCK + threadIdx.x;
    const int g0 = (int)(blockIdx.x * HP_BLOCK);
    const int w0 = g0 - HP_BLOCK;
    for (int k = threadIdx.x; k < HP_BLOCK * 3 + 1; k += HP_BLOCK) { const int j = w0 + k; s_keys[k] = (j >= 0 && j < (int)n) ? skeys[j] : 0u; }
    __syncthreads();
    if (pc >= ni) return;
    auto key_at = [&](int j) -> u64 {
        const u32 k = ((u32)(j - w0) <= (u32)(HP_BLOCK * 3)) ? s_keys[j - w0] : skeys[j];
        return ((u64)k << 32) | (u32)j;
    };
    const int p = (int)pc;
    const u64 kp = key_at(p);
    const int c0 = clz64(kp ^ key_at(p + 1));
    auto inside = [&](int j) -> bool { return j >= 0 && j < (int)n && clz64(key_at(j) ^ kp) >= c0; };
    int step = 1;
    while (inside(p - step)) step <<= 1;
    int lo = p - (step >> 1);
    for (int t = step >> 2; t > 0; t >>= 1) if (inside(lo - t)) lo -= t;
    step = 1;
    while (inside(p + 1 + step)) step <<= 1;
    int hi = p + 1 + (step >> 1);
    for (int t = step >> 2; t > 0; t >>= 1) if (inside(hi + t)) hi += t;
    const u32 size = (u32)(hi - lo + 1);
    const bool big = size > HP_HALF || size == n;
    level_keys[pc] = big ? (u32)(63 - c0) : HP_NO_TASK;
    if (big) ranges[pc] = (u64)(u32)lo | ((u64)(u32)hi << 32);
}

// task records in level order: {gap, L, R, -} — one 16-byte load instead of the dependent task id -> range pair
__global__ __launch_bounds__(256) void k_hp_pack(const u32* __restrict__ level_offsets, const u32* __restrict__ task_ids,
                                                 const u64* __restrict__ ranges, uint4* __restrict__ tasks) {
    const u32 total = level_offsets[62];                      // end of the last level (key 255 = "no task" sorts behind)
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const u32 p = task_ids[i]; const u64 rg = ranges[p];
        tasks[i] = make_uint4(p, (u32)rg, (u32)(rg >> 32), 0u);
    }
}

// level_offsets: the sort's exclusive digit offsets (hist after k_scan_hist): tasks of level v are tasks[off[v] .. off[v+1])
#ifndef HP_LEVEL_WAVES
#define HP_LEVEL_WAVES 1
#endif
__global__ __launch_bounds__(HP_BLOCK, HP_LEVEL_WAVES) void k_hp_level(const u32* __restrict__ level_offsets, int level, const uint4* __restrict__ tasks,
                                                       const bvh_aabb* __restrict__ boxes,
                                                       const u32* __restrict__ svals, bvh_primref* __restrict__ leaves, bvh2_node* nodes,
                                                       u64* cidx, u32* zero_parent, u32 n) {
    const u32 base = level_offsets[level], count = level_offsets[level + 1] - base;
    const int lane = threadIdx.x & (WAVE - 1);
    const int half = lane >> 5, slot = lane & 31, hbase = half << 5;
    const u32 ni = n - 1;
    const u32 stride = gridDim.x * (HP_BLOCK / 32);
    const u32 t0 = blockIdx.x * (HP_BLOCK / 32) + (threadIdx.x >> 6) * 2;          // wave-uniform first task of the wave
    // The merge rounds are short next to the dependent loads in front of them (task record -> cluster entries -> boxes;
    // measured: waves parked on s_waitcnt 66 % of the time), and the compiler serialises loads prefetched across the loop
    // back-edge.  So every iteration carries TWO independent task pairs (A, B): their loads are issued stage by stage
    // together, then the rounds run back to back — twice the memory-level parallelism per wave.
    for (u32 tw = t0; tw < count; tw += 2 * stride) {                                // wave-uniform
        const u32 tA = tw + (u32)half, tB = tw + stride + (u32)half;
        const bool hA = tA < count, hB = tB < count;
        uint4 rA = make_uint4(0, 0, 0, 0), rB = make_uint4(0, 0, 0, 0);
        if (hA) rA = tasks[base + tA];
        if (hB) rB = tasks[base + tB];
        const HpEntry eA = load_entry<false>(hA, rA.y, rA.z, rA.x, svals, cidx, ni, slot);
        const HpEntry eB = load_entry<false>(hB, rB.y, rB.z, rB.x, svals, cidx, ni, slot);
        const HpWork wA = prepare<false>(hA, rA.y, rA.z, eA, boxes, svals, leaves, nodes, ni, slot, hbase);
        const HpWork wB = prepare<false>(hB, rB.y, rB.z, eB, boxes, svals, leaves, nodes, ni, slot, hbase);
        reduce_and_store<false>(wA, nodes, cidx, zero_parent, ni, lane, slot, hbase, 0);
        reduce_and_store<false>(wB, nodes, cidx, zero_parent, ni, lane, slot, hbase, 0);
    }
}

// =====================================================================================================================
// Block-local variant (large inputs).
//
// A workgroup owns T consecutive sorted leaves.  Every LBVH node whose leaf range lies inside those T leaves ("local";
// > 90 % of the merge tasks) is processed from LDS: the block stages its leaves' boxes once (all gathers in flight together —
// this is also SetupClusters), keeps the work lists of its ranges ({id, rep, box} per surviving cluster, at the range's first
// 16 positions exactly like the reference's nodeIdx array) in LDS, and walks its local hierarchy level by level (level =
// 63 - common prefix; a parent's prefix is strictly shorter than its children's) with one barrier per non-empty level.  A
// local task therefore has no global load in front of its PLOC rounds and only the 32-byte node store behind them.
// Nodes whose range crosses a block boundary ("external": the ancestors of the T-aligned gaps) run afterwards under the
// asynchronous dependency protocol of k_hploc (async_climb): the block publishes the survivors of its maximal local ranges
// to global memory and moves the parents' counters.  One launch, no per-level kernel boundaries, no plan/sort passes.
// =====================================================================================================================
template <int T, int NT, int OCC>
__global__ __launch_bounds__(NT, OCC) void k_hploc_block(const bvh_aabb* __restrict__ boxes, const u32* __restrict__ skeys,
                                                    const u32* __restrict__ svals, bvh_primref* __restrict__ leaves,
                                                    bvh2_node* nodes, u64* cidx, u64* ranges, u32* counter, u32* zero_parent,
                                                    u32* queue, u32* queue_count, u32 n, int dbg) {
    constexpr int PER = T / NT;                      // leaf positions (and gaps) per thread
    constexpr int NW = NT / WAVE;
    static_assert(T % NT == 0 && T <= 32768, "block-local HPLOC tile");
    __shared__ u32 s_key[T + 2];                     // sorted keys of positions g0-1 .. g0+T
    // work lists: per position the cluster's id and rep, tile-relative in 16 bits (a cluster merged inside the tile absorbs a
    // partner whose first leaf lies in the tile, so node index = rep' - 1 is tile-local too), and its box (SoA)
    __shared__ unsigned short e_id[T], e_rep[T];     // id: 0x8000 | k = leaf ni + g0 + k;  k = node g0 + k;  0xFFFF = invalid
    __shared__ float e_b[6][T];                      // (later: the block's ready list)
    __shared__ u32 m_range[T];                       // per gap (relative): L | R << 16 of a local big node; M_EXT: range leaves the block
    __shared__ unsigned short s_task[T];             // local big nodes grouped by level
    __shared__ u32 s_cnt[64], s_off[64];
    __shared__ u32 s_nready;

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid >> 6;
    const int half = lane >> 5, slot = lane & 31, hbase = half << 5;
    const u32 ni = n - 1;
    const u32 g0 = blockIdx.x * (u32)T;
    const u32 nleaf = (n - g0) < (u32)T ? (n - g0) : (u32)T;

    // ---- stage the block: leaves (SetupClusters :44-47, fused), keys -------------------------------------------------
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const u32 k = (u32)tid + (u32)i * NT;
        if (k < nleaf) {
            const u32 g = g0 + k, prim = svals[g];
            const Box b = box_load(boxes + prim);
            float* f = reinterpret_cast<float*>(leaves + g);
            reinterpret_cast<u32*>(f)[0] = prim;
            f[1] = b.lx; f[2] = b.ly; f[3] = b.lz; f[4] = b.hx; f[5] = b.hy; f[6] = b.hz;
            e_id[k] = (unsigned short)(0x8000u | k); e_rep[k] = (unsigned short)k;
            e_b[0][k] = b.lx; e_b[1][k] = b.ly; e_b[2][k] = b.lz; e_b[3][k] = b.hx; e_b[4][k] = b.hy; e_b[5][k] = b.hz;
        }
    }
    for (int k = tid; k < T + 2; k += NT) { const long long j = (long long)g0 - 1 + k; s_key[k] = (j >= 0 && j < (long long)n) ? skeys[j] : 0u; }
    if (tid < 64) s_cnt[tid] = 0u;
    if (tid == 0) s_nready = 0u;
    __syncthreads();
    if (dbg == 1) return;

    // ---- ranges of the block's gaps, clamped to the window [g0-1, g0+T]; a range touching the window's rim is external
    constexpr u32 M_EXT = 0xFFFFFFFFu;
    const int jmin = g0 ? (int)g0 - 1 : 0;
    const int jmax = (g0 + (u32)T <= ni) ? (int)(g0 + (u32)T) : (int)ni;
    auto wkey = [&](int j) -> u64 { return ((u64)s_key[j - (int)g0 + 1] << 32) | (u32)j; };
    int my_lv[PER]; u32 my_pos[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const u32 k = (u32)tid + (u32)i * NT;
        const u32 pc = g0 + k;
        my_lv[i] = -1; my_pos[i] = 0;
        if (k < nleaf && pc < ni) {
            const int p = (int)pc;
            const u64 kp = wkey(p);
            const int c0 = clz64(kp ^ wkey(p + 1));
            auto inside = [&](int j) -> bool { return j >= jmin && j <= jmax && clz64(wkey(j) ^ kp) >= c0; };
            int step = 1;
            while (inside(p - step)) step <<= 1;
            int lo = p - (step >> 1);
            for (int t = step >> 2; t > 0; t >>= 1) if (inside(lo - t)) lo -= t;
            step = 1;
            while (inside(p + 1 + step)) step <<= 1;
            int hi = p + 1 + (step >> 1);
            for (int t = step >> 2; t > 0; t >>= 1) if (inside(hi + t)) hi += t;
            const bool ext = lo < (int)g0 || hi > (int)(g0 + (u32)T - 1u);
            m_range[k] = ext ? M_EXT : 0u;
            if (!ext && (u32)(hi - lo + 1) > HP_HALF) {
                m_range[k] = (u32)(lo - (int)g0) | ((u32)(hi - (int)g0) << 16);
                my_lv[i] = 63 - c0;
                my_pos[i] = atomicAdd(&s_cnt[my_lv[i]], 1u);
            }
        }
    }
    __syncthreads();
    if (tid < 64) {                                  // exclusive scan of the level counts
        const u32 v = s_cnt[tid]; u32 incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = (u32)__shfl_up((int)incl, d); if (lane >= d) incl += t; }
        s_off[tid] = incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) if (my_lv[i] >= 0) s_task[s_off[my_lv[i]] + my_pos[i]] = (unsigned short)((u32)tid + (u32)i * NT);
    __syncthreads();

    if (dbg == 2) return;
    // ---- local hierarchy, deepest level first; two tasks per wave pass ----------------------------------------------------
    for (int lv = 0; lv < 62; ++lv) {
        const u32 c = s_cnt[lv];
        if (!c) continue;                            // block-uniform
        const u32 base = s_off[lv];
        for (u32 tw = (u32)wave * 2u; tw < c; tw += (u32)NW * 2u) {
            const u32 t = tw + (u32)half;
            const bool have = t < c;
            u32 P = 0, L = 0, R = 0;
            if (have) { P = s_task[base + t]; const u32 rg = m_range[P]; L = rg & 0xFFFFu; R = rg >> 16; }
            // loadIndices (:192-206) from the LDS work lists: the first <= 16 valid entries of each child range
            const bool is_left = slot < 16;
            const u32 kk = (u32)(slot & 15);
            const u32 c_start = is_left ? L : P + 1u, c_len = is_left ? (P - L + 1u) : (R - P);
            u32 idv = 0xFFFFu;
            if (have && kk < c_len) idv = e_id[c_start + kk];
            const u32 vb = (u32)(__ballot(idv != 0xFFFFu) >> hbase);
            const u32 nl = (u32)__popc(vb & 0xFFFFu), nr = (u32)__popc(vb >> 16);
            HpWork w; w.have = have; w.final_ = false; w.tL = g0 + L; w.cnt = nl + nr;
            w.id = INV; w.rep = INV; w.b = box_empty();
            if (have && (u32)slot < w.cnt) {
                const u32 sp = (u32)slot < nl ? L + (u32)slot : P + 1u + ((u32)slot - nl);
                const u32 ie = e_id[sp];
                w.id = (ie & 0x8000u) ? ni + g0 + (ie & 0x7FFFu) : g0 + ie; w.rep = g0 + e_rep[sp];
                w.b = { e_b[0][sp], e_b[1][sp], e_b[2][sp], e_b[3][sp], e_b[4][sp], e_b[5][sp] };
            }
            ploc_rounds<true>(w, nodes, zero_parent, ni, lane, slot, hbase, 0);
            if (have && slot < 16) {                 // storeIndices (:208-218) into the range's first 16 positions
                const u32 d = L + (u32)slot;
                e_id[d] = (unsigned short)(w.id == INV ? 0xFFFFu : (w.id >= ni ? 0x8000u | (w.id - ni - g0) : w.id - g0));
                e_rep[d] = (unsigned short)(w.rep - g0);
                e_b[0][d] = w.b.lx; e_b[1][d] = w.b.ly; e_b[2][d] = w.b.lz; e_b[3][d] = w.b.hx; e_b[4][d] = w.b.hy; e_b[5][d] = w.b.hz;
            }
        }
        __syncthreads();
    }
    drain_stores();                                  // every wave's node stores are in memory before the block publishes
    __syncthreads();
    if (dbg == 3) return;

    // ---- hand-over: survivors of the maximal local ranges go to global memory and move their (external) parent's count; the
    // block's external nodes contribute what their own thread knows (which children are small, and those children's far ends).
    // Nodes whose count completes here are queued for k_hploc_ext.
    auto gkey = [&](int j) -> u64 {
        const u32 kv = (j >= jmin && j <= jmax) ? s_key[j - (int)g0 + 1] : skeys[j];
        return ((u64)kv << 32) | (u32)j;
    };
    bool ev[PER]; u32 ev_pc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const u32 k = (u32)tid + (u32)i * NT;
        const u32 pc = g0 + k;
        ev[i] = false; ev_pc[i] = 0;
        if (k < nleaf && pc < ni) {
            u32* rq = reinterpret_cast<u32*>(ranges + pc);
            if (m_range[k] == M_EXT) {
                const int p = (int)pc;
                const u64 kp = gkey(p);
                const int c0 = clz64(kp ^ gkey(p + 1));
                auto inside = [&](int j) -> bool { return j >= 0 && j < (int)n && clz64(gkey(j) ^ kp) >= c0; };
                // child [L, p] is big iff leaf p-16 is inside; child [p+1, R] is big iff leaf p+17 is inside
                const bool lbig = inside(p - (int)HP_HALF), rbig = inside(p + 1 + (int)HP_HALF);
                int lo = p, hi = p + 1;
                if (!lbig) { for (int t = 8; t > 0; t >>= 1) if (inside(lo - t)) lo -= t; }          // L in [p-15, p]
                if (!rbig) { for (int t = 8; t > 0; t >>= 1) if (inside(hi + t)) hi += t; }          // R in [p+1, p+16]
                const u32 e = (lbig ? 1u : 0u) + (rbig ? 1u : 0u);
                if (e == 0u) {
                    if ((u32)(hi - lo + 1) > HP_HALF) { rq[0] = (u32)lo; rq[1] = (u32)hi; ev[i] = true; ev_pc[i] = pc; }   // read by the next launch
                } else {
                    if (!lbig) st_agent(rq + 0, (u32)lo);
                    if (!rbig) st_agent(rq + 1, (u32)hi);
                    drain_stores();
                    const u32 old = __hip_atomic_fetch_add(counter + pc, 3u - e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((old + (3u - e)) == 3u) { ev[i] = true; ev_pc[i] = pc; }
                }
            } else if (my_lv[i] >= 0) {
                const u32 rg = m_range[k];
                const u32 Lr = rg & 0xFFFFu, L = g0 + Lr, R = g0 + (rg >> 16);
                u32 q;                                                               // findParent (:66-81)
                if (L == 0u) q = R;
                else if (R == ni) q = L - 1u;
                else q = ((wkey((int)R) ^ wkey((int)R + 1)) < (wkey((int)L - 1) ^ wkey((int)L))) ? R : L - 1u;
                if (q < g0 || m_range[q - g0] == M_EXT) {
#pragma unroll
                    for (int sidx = 0; sidx < 16; ++sidx) {
                        const u32 ie = e_id[Lr + sidx];
                        const u32 idg = ie == 0xFFFFu ? INV : ((ie & 0x8000u) ? ni + g0 + (ie & 0x7FFFu) : g0 + ie);
                        st_agent(cidx + L + sidx, entry(idg, g0 + e_rep[Lr + sidx]));
                    }
                    st_agent(reinterpret_cast<u32*>(ranges + q) + (q == R ? 0 : 1), q == R ? L : R);
                    drain_stores();
                    const u32 old = __hip_atomic_fetch_add(counter + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old + 1u == 3u) { ev[i] = true; ev_pc[i] = q; }
                }
            }
        }
    }
    __syncthreads();                                 // the work lists are dead: their LDS becomes the block's ready list
    u32* r_pc = reinterpret_cast<u32*>(&e_b[0][0]);
#pragma unroll
    for (int i = 0; i < PER; ++i) if (ev[i]) r_pc[atomicAdd(&s_nready, 1u)] = ev_pc[i];
    __syncthreads();
    const u32 nready = s_nready;                     // <= T (one event per gap at most)
    if (dbg == 4) return;
    if (nready) {
        if (tid == 0) s_off[0] = atomicAdd(queue_count, nready);
        __syncthreads();
        const u32 at = s_off[0];
        for (u32 i = (u32)tid; i < nready; i += (u32)NT) queue[at + i] = r_pc[i];
    }
}

// External nodes (ranges crossing the tiles of k_hploc_block): the queue holds the nodes whose dependencies were complete when
// the block kernel ended; every wave takes two at a time and climbs while it keeps completing parents (async_climb).
__global__ __launch_bounds__(256) void k_hploc_ext(const bvh_aabb* __restrict__ boxes, const u32* __restrict__ skeys,
                                                   const u32* __restrict__ svals, bvh_primref* __restrict__ leaves, bvh2_node* nodes,
                                                   u64* cidx, u64* ranges, u32* counter, u32* zero_parent,
                                                   const u32* __restrict__ queue, const u32* __restrict__ queue_count, u32 n) {
    const int lane = threadIdx.x & (WAVE - 1);
    const u32 total = *queue_count;
    const u32 nwaves = gridDim.x * (256 / WAVE);
    const u32 wid = blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
    for (u32 base = wid * 2u; base < total; base += nwaves * 2u) {      // wave-uniform
        const u32 idx = base + (u32)(lane >> 5);
        const bool ready = (lane & 31) == 0 && idx < total;
        u32 pc = 0, L = 0, R = 0;
        if (ready) { pc = queue[idx]; const u64 rg = ranges[pc]; L = (u32)rg; R = (u32)(rg >> 32); }
        async_climb<false, true>(ready, pc, L, R, boxes, skeys, svals, leaves, nodes, cidx, ranges, counter, zero_parent, n - 1, lane, 0);
    }
}

void launch_hploc(hipStream_t s, const void* d_boxes, const uint32_t* d_skeys, const uint32_t* d_svals, uint32_t n,
                  void* d_nodes, void* d_leaves, uint64_t* d_cluster_idx, uint64_t* d_ranges, uint32_t* d_counter, uint32_t* d_zero_parent) {
    hipMemsetAsync(d_counter, 0, (size_t)n * sizeof(u32), s);
    const u32 gaps = n - 1;
    { KernelScope ks(s, "k_hploc"); hipLaunchKernelGGL(k_hploc, dim3((gaps + HP_BLOCK - 1) / HP_BLOCK), dim3(HP_BLOCK), 0, s, (const bvh_aabb*)d_boxes, d_skeys, d_svals,
                       (bvh_primref*)d_leaves, (bvh2_node*)d_nodes, d_cluster_idx, d_ranges, d_counter, d_zero_parent, n, hploc_ablation()); }
}

// Block-local HPLOC for large n (n > 2 tiles: the root is never local).  queue: u32[n] scratch; queue_count: one word.
#ifndef HPB_T
#define HPB_T 1024
#endif
#ifndef HPB_NT
#define HPB_NT 512
#endif
#ifndef HPB_OCC
#define HPB_OCC 8
#endif
static void hpb_config(int* t, int* nt, int* occ) {
    *t = HPB_T; *nt = HPB_NT; *occ = HPB_OCC;
    const char* e = getenv("BVH_HPB");                 // "T,NT,OCC" (measurements only)
    if (e) { int a = 0, b = 0, c = 0; if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) { *t = a; *nt = b; *occ = c; } }
}
uint32_t hploc_block_tile() { int t, nt, occ; hpb_config(&t, &nt, &occ); return (uint32_t)t; }
void launch_hploc_block(hipStream_t s, const void* d_boxes, const uint32_t* d_skeys, const uint32_t* d_svals, uint32_t n,
                        void* d_nodes, void* d_leaves, uint64_t* d_cluster_idx, uint64_t* d_ranges, uint32_t* d_counter, uint32_t* d_zero_parent,
                        uint32_t* d_queue, uint32_t* d_queue_count) {
    (void)hipMemsetAsync(d_counter, 0, (size_t)n * sizeof(u32), s);
    (void)hipMemsetAsync(d_queue_count, 0, sizeof(u32), s);
    int t, nt, occ; hpb_config(&t, &nt, &occ);
    const int dbg = hploc_ablation();
#define HPB_LAUNCH(TT, NN, OO) hipLaunchKernelGGL((k_hploc_block<TT, NN, OO>), dim3((n + TT - 1) / TT), dim3(NN), 0, s, (const bvh_aabb*)d_boxes, d_skeys, d_svals, \
                       (bvh_primref*)d_leaves, (bvh2_node*)d_nodes, d_cluster_idx, d_ranges, d_counter, d_zero_parent, d_queue, d_queue_count, n, dbg)
    { KernelScope ks(s, "k_hploc_block");
      if (t == 1024 && nt == 512 && occ == 8) HPB_LAUNCH(1024, 512, 8);
      else if (t == 512 && nt == 256 && occ == 8) HPB_LAUNCH(512, 256, 8);
      else if (t == 512 && nt == 256 && occ == 1) HPB_LAUNCH(512, 256, 1);
      else if (t == 2048 && nt == 1024 && occ == 8) HPB_LAUNCH(2048, 1024, 8);
      else if (t == 512 && nt == 128 && occ == 8) HPB_LAUNCH(512, 128, 8);
      else if (t == 256 && nt == 128 && occ == 8) HPB_LAUNCH(256, 128, 8);
      else if (occ == 1) HPB_LAUNCH(1024, 512, 1);
      else HPB_LAUNCH(1024, 512, 8); }
#undef HPB_LAUNCH
    if (dbg) return;
    KernelScope ks(s, "k_hploc_ext");
    hipLaunchKernelGGL(k_hploc_ext, dim3(2048), dim3(256), 0, s, (const bvh_aabb*)d_boxes, d_skeys, d_svals, (bvh_primref*)d_leaves, (bvh2_node*)d_nodes,
                       d_cluster_idx, d_ranges, d_counter, d_zero_parent, (const u32*)d_queue, (const u32*)d_queue_count, n);
}

// Level-synchronous HPLOC for large n.  level_keys / task_keys / task_ids: u32[n] scratch; sc: the sort's scratch (re-armed here).
void launch_hploc_levels(hipStream_t s, const SortScratch& sc, const void* d_boxes, const uint32_t* d_skeys, const uint32_t* d_svals, uint32_t n,
                         void* d_nodes, void* d_leaves, uint64_t* d_cluster_idx, uint64_t* d_ranges, uint32_t* d_level_keys,
                         uint32_t* d_task_keys, uint32_t* d_task_ids, uint4* d_tasks, uint32_t* d_zero_parent) {
    const u32 gaps = n - 1;
    { KernelScope ks(s, "k_hp_plan"); hipLaunchKernelGGL(k_hp_plan, dim3((gaps + HP_BLOCK - 1) / HP_BLOCK), dim3(HP_BLOCK), 0, s, d_skeys, d_ranges, d_level_keys, n); }
    sort_prepare(s, sc, gaps);
    sort_pairs(s, sc, d_level_keys, nullptr, gaps, d_task_keys, d_task_ids, 0, 8, false);     // one pass; sc.hist = level offsets
    hipLaunchKernelGGL(k_hp_pack, dim3(1024), dim3(256), 0, s, (const u32*)sc.hist, (const u32*)d_task_ids, (const u64*)d_ranges, d_tasks);
    const u32 max_tasks = gaps / 17 + 1;
    u32 grid = (max_tasks + (HP_BLOCK / 32) - 1) / (HP_BLOCK / 32);
    if (grid > 2048u) grid = 2048u;
    KernelScope ks(s, "k_hp_level");                  // all 62 launches are timed as one group
    for (int level = 0; level < 62; ++level) {       // level = 63 - c0, c0 in [2, 63]
        hipLaunchKernelGGL(k_hp_level, dim3(grid), dim3(HP_BLOCK), 0, s, (const u32*)sc.hist, level, (const uint4*)d_tasks,
                           (const bvh_aabb*)d_boxes, d_svals, (bvh_primref*)d_leaves, (bvh2_node*)d_nodes, d_cluster_idx, d_zero_parent, n);
    }
}

} // namespace bvh
