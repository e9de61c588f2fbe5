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
//  * a merge task occupies one 32-lane half of a wave64: the work list (id, rep, box) lives in registers, neighbours come
//    from DPP wave shifts, partners through ds_bpermute, compaction through ds_permute.  Two tasks run side by side in the
//    two halves.  No barriers inside a task, no reliance on store conflict order.  (The tile kernel keeps its
//    lists in LDS instead — ploc_rounds_lds — on an interleaved lane layout whose neighbour boxes are row-shift DPP operands: nn_search_il.)
//  * large inputs (k_hploc_block): a workgroup owns a tile of T consecutive sorted leaves and processes every node whose
//    range lies inside the tile out of LDS, level by level; only the nodes that cross tile boundaries (the ancestors of the
//    T-aligned gaps, ~15 % of the tasks at T = 512) use the inter-workgroup protocol below, in a second launch (k_hploc_ext).
//    Small inputs (k_hploc): one launch, every big node uses the protocol.
//  * inter-workgroup protocol: ONE 64-bit word per node, dep[p] = {count:2 | R:30 | L:30}.  Only "big" nodes (range > 16)
//    take part.  A finished left child adds {1, 0, its L}, a finished right child {1, its R, 0}; the node's own thread adds
//    {3 - #big children, far ends of its small children}.  Whoever brings the count to 3 has the node's full range in the sum,
//    resets the word to 0 (so the array is clean for the next build — no memset per build), runs the node's merge task and
//    climbs.  Nobody waits, nobody searches for a range, and the range travels inside the atomic that publishes the child.
//  * survivors of a finished range are published as 32-byte records {id, rep, box} at the range's first 16 positions (the
//    reference's nodeIdx array holds ids only and every consumer gathers the boxes again): one dependent load per task.
//  * SetupClusters is fused (a leaf's PrimRef is written when the leaf is first staged, exactly once).
//  * node allocation: the reference takes node indices from ONE global counter (:163-167); a single word sustains ~90
//    returning atomics/us on MI355X (~23 ms for a 10 M build).  Here every cluster carries the sorted position of its first
//    leaf ("rep"); lists stay ordered by rep, a merge keeps the lower partner's rep and retires the absorbed partner's rep
//    r in [1,n) exactly once — node index r-1 is a bijection onto [0,n-1).  The final merge moves whatever occupies node 0 to
//    its own natural slot so that the root is node 0, as the reference guarantees.  Numbering depends on the topology only.
//
// Hand-off between waves (possibly on different XCDs): records and node boxes written during a launch are agent-scope
// write-through stores read back with agent-scope loads; a wave drains its stores (s_waitcnt vmcnt(0)) before the agent-scope
// atomic on dep[] that publishes a finished range.
#include <cstdlib>
#include <cstdio>
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

// ablation switch for measurements only (results are incomplete when set): compiled in with -DBVH_ABLATION (tools/build_variant.sh)
static inline int hploc_ablation() {
#ifdef BVH_ABLATION
    const char* e = getenv("BVH_HPLOC_DEBUG"); return e ? atoi(e) : 0;
#else
    return 0;
#endif
}

constexpr int HP_BLOCK = 256;
constexpr u32 HP_HALF = 16;        // WarpSize/2 of the reference's wave32 (src/HplocKernel.h:195,238)
constexpr int HP_RADIUS = 8;       // PlocRadius, src/Common.h:595

__device__ __forceinline__ int clz64(u64 v) { return v ? __clzll((long long)v) : 64; }
// push semantics: lane l's value lands in lane dst(l)
__device__ __forceinline__ u32 push_u32(int dst, u32 v) { return (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)v); }
__device__ __forceinline__ float push_f32(int dst, float v) { return __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(v))); }

// ---- dependency words -------------------------------------------------------------------------------------------------
constexpr u64 DEP_MASK = (1ull << 30) - 1ull;
__device__ __forceinline__ u64 dep_word(u32 count, u32 L, u32 R) { return ((u64)count << 60) | ((u64)R << 30) | (u64)L; }
// add `mine` to dep[q]; true when that completes the node (count 3): then [L,R] is its range and the word is clean again
__device__ __forceinline__ bool dep_arrive(u64* dep, u32 q, u64 mine, u32& L, u32& R) {
    const u64 tot = __hip_atomic_fetch_add(dep + q, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine;
    if ((tot >> 60) != 3ull) return false;
    st_agent(dep + q, 0ull);                       // nobody touches a completed node's word again during this build
    L = (u32)(tot & DEP_MASK); R = (u32)((tot >> 30) & DEP_MASK);
    return true;
}

// findParent (:66-81): the boundary gap with the longer common prefix (smaller xor) is the parent of range [L,R] (not the root)
template <typename Closer>
__device__ __forceinline__ u32 parent_gap(u32 L, u32 R, u32 ni, Closer pair_closer) {   // pair_closer(a, b): is (a, a+1) closer than (b, b+1)?
    if (L == 0u) return R;
    if (R == ni) return L - 1u;
    return pair_closer(R, L - 1u) ? R : L - 1u;
}

// Survivor records of a finished range [L, R] (> 16 leaves): sixteen 32-byte records inside the range's own positions of `recs`.  Round 4: a range that is the LEFT
// child of its parent (parent gap == R) stores them at its LAST sixteen positions, a right child at its first sixteen — both children's records then sit around the
// parent's split p ([p - 15, p] and [p + 1, p + 16]) and are addressable from p alone, before the parent's range is known (k_hploc_ext's look-ahead).
__device__ __forceinline__ u32 rec_base(bool left_child, u32 L, u32 R) { return left_child ? R - (HP_HALF - 1u) : L; }

struct HpWork { u32 id, rep, cnt, tL; Box b; bool have, final_; };

// loadIndices (:192-206) + the box fetch of plocMerge (:242-246): slots 0..15 <- left child [tL, tP], 16..31 <- right child
// [tP+1, tR]; a child of > 16 leaves left its survivors as records, a smaller one is still its leaves; left-packed result.
// SETUP: this is the first touch of those leaves — fused SetupClusters (:44-47); otherwise their PrimRefs exist already.
template <bool SETUP>
__device__ __forceinline__ HpWork load_work(bool have, u32 tL, u32 tR, u32 tP, const bvh_aabb* __restrict__ boxes, const u32* __restrict__ svals,
                                            bvh_primref* leaves, const bvh2_node* recs, u32 ni, int slot, int hbase) {
    const bool is_left = slot < 16;
    const u32 s = (u32)(slot & 15);
    const u32 c_start = is_left ? tL : tP + 1u, c_len = is_left ? (tP - tL + 1u) : (tR - tP);
    u32 id = INV, rep = INV;
    Box b = box_empty();
    const bool leaf = have && c_len <= HP_HALF && s < c_len;
    if (leaf) {
        rep = c_start + s; id = ni + rep;
        if (SETUP) {
            const u32 prim = svals[rep];
            b = box_gather(boxes + prim);
            float* f = reinterpret_cast<float*>(leaves + rep);
            reinterpret_cast<u32*>(f)[0] = prim;
            f[1] = b.lx; f[2] = b.ly; f[3] = b.lz; f[4] = b.hx; f[5] = b.hy; f[6] = b.hz;
        } else b = box_load_u(reinterpret_cast<const bvh_aabb*>(reinterpret_cast<const float*>(leaves + rep) + 1));
    }
    if (have && c_len > HP_HALF) rec_load_agent(recs + rec_base(is_left, c_start, tP) + s, id, rep, b);     // (left child [tL, tP]: right-aligned at tP; right child: at tP + 1)
    // left-pack: slot t < nl <- slot t ; slot t in [nl, cnt) <- slot 16 + (t - nl)
    const u32 vb = (u32)(__ballot(id != INV) >> hbase);
    const u32 nl = (u32)__popc(vb & 0xFFFFu), nr = (u32)__popc(vb & 0xFFFF0000u);
    HpWork w; w.cnt = nl + nr; w.tL = tL; w.have = have; w.final_ = have && tL == 0 && tR == ni;
    const int src = hbase + (((u32)slot < nl) ? slot : (int)((16 + (u32)slot - nl) & 31));
    const u32 ti = (u32)__shfl((int)id, src);
    w.rep = (u32)__shfl((int)rep, src);
    w.b = shfl_box(b, src);
    w.id = ((u32)slot < w.cnt) ? ti : INV;
    return w;
}

#ifndef HP_NN_LDS
#define HP_NN_LDS 3        // neighbour selection.  3 (default since round 4): the pair's key is minimised into the FAR end's word by an LDS atomic (the reference's
                           //    formulation) and into the lane's own running minimum by ONE v_min_f64 on the same 64-bit key (eight LDS atomics per round instead of
                           //    sixteen; needs -fno-slp-vectorize, see the Makefile); 1: both ends by LDS atomics (rounds 1-3; the alt variant of tests/test_gpu_variants.py).
                           //    (2 = own end as a compare-select chain and 0 = two running minima + ds_bpermute were measured slower: LEADS.md, tools/probes/r05_pruned_switches.patch)
#endif
// findNearestNeighbours (:83-117) of one PLOC round for the two tasks of a wave: every pair (slot, slot + r), r = 1..8, is evaluated ONCE, by its lower end —
// neighbour boxes arrive through a DPP wave_shl:1 chain, one candidate per step in scalar f32 (a packed f32 operation costs what two scalar ones do on gfx950 and the
// two-candidate form kept twelve more registers alive) — and its 64-bit key {area bits, other end's slot} is minimised into BOTH ends: the far end's word with an LDS
// atomic (ds_min_u64: a wave's LDS operations execute in order, so the reset, the atomics and the read-back need no barrier), the lane's own minimum in a register
// pair (HP_NN_LDS = 3; = 1: an LDS atomic too).  Returns the slot of the lane's nearest neighbour (lowest slot on equal areas).  PUBLISH: the choice is also left in
// the low half of the lane's key word (ploc_rounds_lds reads it there).  nn: the wave's 64-entry LDS scratch.  ABL_EXTRA_*: in-situ cost probes (wrong trees, timing
// only; profiles/r04_tile_phases.md).
template <bool PUBLISH = false>
__device__ __forceinline__ int nn_search(const Box& b, bool act, u32 cnt, int lane, int slot, u64* nn) {
    nn[lane] = ~0ull;
    compiler_fence();                        // reset, atomics and read-back stay in program order
#if HP_NN_LDS == 3
    // the lane's own right-hand candidates: the SAME 64-bit key {area bits, other end's slot}, minimised in a register pair by v_min_f64 — one VALU instruction
    // instead of one LDS atomic.  An area is a non-negative f32, so the key read as an f64 is a non-negative finite number (its exponent field is the area's
    // sign, exponent and three mantissa bits: never all ones), and non-negative doubles order like their bit patterns; f64 denormals (areas below 2^-126) are
    // compared, not flushed (the f64 denormal mode of every HIP kernel is "preserve"), and a minimum returns one of its operands bit for bit.
    double own = __longlong_as_double(0x7FEFFFFFFFFFFFFFll);
#endif
    Box nb = b;
#pragma unroll
    for (int rr = 1; rr <= HP_RADIUS; ++rr) {
        nb = box_shl1(nb);                                                       // box of slot + rr
        const float ex = fmaxf(nb.hx, b.hx) - fminf(nb.lx, b.lx), ey = fmaxf(nb.hy, b.hy) - fminf(nb.ly, b.ly), ez = fmaxf(nb.hz, b.hz) - fminf(nb.lz, b.lz);
        const float half_area = ex * ey + ex * ez + ey * ez;                     // Aabb::area (:361-365): 2 * (xy + xz + yz)
        const u32 ab = __float_as_uint(half_area + half_area);                   // (x + x == 2 * x exactly)
        if (act && (u32)(slot + rr) < cnt) {                                     // both ends are clusters of this task
            atomicMin(reinterpret_cast<unsigned long long*>(nn + lane + rr), ((unsigned long long)ab << 32) | (u32)slot);
#if HP_NN_LDS == 3
            own = __builtin_fmin(own, __longlong_as_double((long long)(((unsigned long long)ab << 32) | (u32)(slot + rr))));
#else
            atomicMin(reinterpret_cast<unsigned long long*>(nn + lane), ((unsigned long long)ab << 32) | (u32)(slot + rr));
#endif
        }
    }
    int probe = 0;
#ifdef ABL_EXTRA_BPERM   // 8 more LDS crossbar operations per round
#pragma unroll
    for (int e = 0; e < 8; ++e) probe ^= __builtin_amdgcn_ds_bpermute(((lane + e) & 63) << 2, (int)__float_as_uint(b.lx) + e);
#endif
#ifdef ABL_EXTRA_VALU    // 192 more VALU operations per round (a dependent chain of v_mul, v_mov_dpp, v_add, v_min)
    { float acc = b.lx;
#pragma unroll
      for (int e = 0; e < 48; ++e) acc = fminf(acc * 1.0000001f, dpp_shl1(acc) + b.ly);
      probe ^= (int)__float_as_uint(acc); }
#endif
#if HP_NN_LDS == 1
    compiler_fence();
    return (int)(u32)nn[lane] | (probe == 0x7fffabcd ? 64 : 0);
#elif HP_NN_LDS == 3
    compiler_fence();
    const u64 left = nn[lane];               // minimum over the pairs (slot - r, slot), or all ones
    const u64 right = (u64)__double_as_longlong(own);
    const u32 choice = (u32)(left < right ? left : right);
    if (PUBLISH) {                           // (ploc_rounds_lds reads the neighbour's choice from the low half of its key word)
        reinterpret_cast<u32*>(nn + lane)[0] = choice;
        compiler_fence();
    }
    return (int)choice | (probe == 0x7fffabcd ? 64 : 0);
#else
#error "HP_NN_LDS is 3 (default) or 1"
#endif
}

// The same search with ONE task on the whole wave (tile kernel, HPB_WIDE: the rounds in which only one half of the wave still has a task — a pass that got a single task,
// or the longer-running task of an unequal pair: 19 of a tile's 56 wave-rounds on the 10 M uniform mesh, tools/model_hploc.py's task trees).  The task's own half evaluates the
// pairs (slot, slot + 1..4), the other half — whose lanes hold the same 32 clusters, read from the task's LDS list — the pairs (slot, slot + 5..8): four candidates per lane
// instead of eight, the keys meet in the task's 32 words of nn through the same LDS minima (tie rule unchanged: the key carries the other end's slot).
// b: the cluster of this lane's slot; nb: the cluster of slot + off (off = 0 in the task's half, 4 in the helping half); ai = first word of the task's half + slot.
__device__ __forceinline__ int nn_search_wide(const Box& b, Box nb, u32 off, u32 cnt, int ai, int slot, int lane, u64* nn) {
    nn[lane] = ~0ull;
    compiler_fence();
#pragma unroll
    for (int rr = 1; rr <= HP_RADIUS / 2; ++rr) {
        nb = box_shl1(nb);                                                       // box of slot + off + rr (a shift across the wave's middle only feeds masked pairs)
        const float ex = fmaxf(nb.hx, b.hx) - fminf(nb.lx, b.lx), ey = fmaxf(nb.hy, b.hy) - fminf(nb.ly, b.ly), ez = fmaxf(nb.hz, b.hz) - fminf(nb.lz, b.lz);
        const float half_area = ex * ey + ex * ez + ey * ez;
        const u32 ab = __float_as_uint(half_area + half_area);
        const u32 far = (u32)slot + off + (u32)rr;
        if (far < cnt) {
            atomicMin(reinterpret_cast<unsigned long long*>(nn + ai + (int)off + rr), ((unsigned long long)ab << 32) | (u32)slot);
            atomicMin(reinterpret_cast<unsigned long long*>(nn + ai), ((unsigned long long)ab << 32) | far);
        }
    }
    compiler_fence();
    return (int)(u32)nn[lane];
}

// The same search for the tile kernel's INTERLEAVED lane layout (HPB_IL): lanes 0..15 of a half hold the task's even slots, lanes 16..31 the odd ones, and every lane
// also reads o = the box of slot + 1 from the LDS list.  The box of slot + r is then a 16-lane ROW shift of b (r even: by r / 2) or of o (r odd: by (r - 1) / 2; r = 1: o
// itself) — a DPP operand of the union's v_min / v_max, no data movement: the 48 v_mov_b32_dpp of the wave_shl chain are gone.  A shift that leaves the row reads 0; such
// a pair has slot + r >= 32 >= cnt and is masked.  No branch around a candidate (the compiler would sink the union into it, and a DPP operand cannot follow a narrowed
// EXEC mask): a pair that does not exist carries the largest finite f64 exponent word instead of its area — it never wins a minimum, neither the far end's LDS word (which
// may be a word of the wave's other half, or one of the 8 pad words behind the last wave's) nor the lane's own v_min_f64.  nnh: the half's key words, BY SLOT.
template <typename List>
__device__ __forceinline__ int nn_search_il(const Box& b, const List& list, u32 opos, bool act, u32 cnt, int slot, u64* nnh) {
    nnh[slot] = ~0ull;
    const Box o = list.load_box(opos);
    compiler_fence();
    double own = __longlong_as_double(0x7FEFFFFFFFFFFFFFll);
#define HP_IL_CAND(RR, SRC, K) { \
        const Box n_ = (K) == 0 ? (SRC) : row_shl<((K) == 0 ? 1 : (K))>(SRC); \
        const float ex = fmaxf(n_.hx, b.hx) - fminf(n_.lx, b.lx), ey = fmaxf(n_.hy, b.hy) - fminf(n_.ly, b.ly), ez = fmaxf(n_.hz, b.hz) - fminf(n_.lz, b.lz); \
        const float half_area = ex * ey + ex * ez + ey * ez; \
        const u32 ab = (act && (u32)(slot + (RR)) < cnt) ? __float_as_uint(half_area + half_area) : 0x7FEFFFFFu; \
        atomicMin(reinterpret_cast<unsigned long long*>(nnh + slot + (RR)), ((unsigned long long)ab << 32) | (u32)slot); \
        own = __builtin_fmin(own, __longlong_as_double((long long)(((unsigned long long)ab << 32) | (u32)(slot + (RR))))); }
    HP_IL_CAND(2, b, 1) HP_IL_CAND(4, b, 2) HP_IL_CAND(6, b, 3) HP_IL_CAND(8, b, 4)
    HP_IL_CAND(1, o, 0) HP_IL_CAND(3, o, 1) HP_IL_CAND(5, o, 2) HP_IL_CAND(7, o, 3)
#undef HP_IL_CAND
    compiler_fence();
    const u64 left = nnh[slot];
    const u64 right = (u64)__double_as_longlong(own);
    const u32 choice = (u32)(left < right ? left : right);
    // (ploc_rounds_lds reads the neighbour's choice from the low half of its key word.)  The WHOLE word is written, high half 0: the unmasked far-end atomics of the
    // previous wave's upper half reach this wave's slots 0..7, unordered with this wave — an all-but-largest key arriving after the choice was left here must not win
    nnh[slot] = (u64)choice;
    compiler_fence();
    return (int)choice;
}

// PLOC rounds (findNearestNeighbours + mergeClusters) until <= 16 clusters (root: 1) remain; the work list stays in
// registers (w is updated in place).  AGENT: node stores are agent-scope write-through because other workgroups of the SAME launch
// read them; the block kernel's nodes are only read by later launches and use plain (cached, write-combined) stores.
// nn: the wave's 64-entry LDS scratch for the nearest-neighbour keys (nn_search)
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// hook: called at the end of every round (the measurement builds count rounds there; tools/probes/hploc_ext_lookahead_wide.patch consumed an early load)
template <bool AGENT = true, typename Hook = NoHook>
__device__ __forceinline__ void ploc_rounds(HpWork& w, bvh2_node* nodes, u32* zero_parent, u32 ni, int lane, int slot, int hbase, u64* nn, Hook hook = Hook()) {
        const bool have = w.have, final_ = w.final_;
        u32 id = w.id, rep = w.rep, cnt = w.cnt;
        Box b = w.b;
        const u32 threshold = final_ ? 1u : HP_HALF;
        while (__ballot(have && cnt > threshold)) {
            const bool act = have && cnt > threshold;
            const int nbr = nn_search(b, act, cnt, lane, slot, nn);
            // mergeClusters (:126-190)
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
                    const u64 w0 = ld_agent(q0 + 0), w1 = ld_agent(q0 + 1), w2 = ld_agent(q0 + 2), w3 = ld_agent(q0 + 3);
                    st_agent(qs + 0, w0); st_agent(qs + 1, w1); st_agent(qs + 2, w2); st_agent(qs + 3, w3);
                    if (l == 0u) l = at;
                    else if (r == 0u) r = at;
                    else {
                        const u32 pw = ld_agent(zero_parent);
                        st_agent(reinterpret_cast<u32*>(nodes + (pw >> 1)) + (pw & 1u), at);
                    }
                    at = 0u;
                } else if (l == 0u || r == 0u) st_agent(zero_parent, (at << 1) | (r == 0u ? 1u : 0u));   // who points at node 0
                if (AGENT) node_store_agent(nodes + at, l, r, b); else node_store_plain(nodes + at, l, r, b);
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
            hook();
        }
        w.id = id; w.rep = rep; w.cnt = cnt; w.b = b;
}

// ---- PLOC rounds on a work list that lives in LDS (round 3) -----------------------------------------------------------------------------
// What binds the tile kernel is the LDS pipeline, not VALU issue (profiles/r03_hploc_bound.md: dropping all 42 DPP moves of a round changes nothing,
// eight more ds_bpermute per round cost +32 %): a ds_bpermute / ds_permute / ds_min_u64 occupies the CU's LDS pipe for 6 cycles per wave-instruction
// whatever it moves (4 bytes per lane for the crossbar operations), a ds_read_b64 for 2.  ploc_rounds above moves the partner's cluster (9
// ds_bpermute) and compacts the list (8 ds_permute) through the crossbar: 102 of a round's ~206 LDS cycles.  Here the list stays in LDS as records:
// the partner is READ (one 4-byte + three 8-byte reads: 10 cycles), survivors are WRITTEN to their rank (22) and every lane reads its new slot
// back (10), the neighbour's choice is a 4-byte read of its key word instead of a ds_bpermute: ~150 cycles per round, and a task neither loads nor
// stores its list around the rounds — the survivors already sit at the range's first positions, where the parent task expects them.
// List: position -> record {id, rep, box}; positions of the task's slots: slot t < nl at base + t, the others at rbase + t (the two children's lists,
// left-packed on the fly); after the first round everything is at base + t.
// Tag: the list's own encoding of {id, rep} (the rounds only decode it where a node is written).  Every list access is unconditional (positions are
// clamped by the caller) and branch-free: a divergent branch between two LDS reads makes them two dependent round trips.
struct TileList {            // k_hploc_block: id / rep tile-relative in 16 bits (id: 0x8000 | k = leaf ni + g0 + k; k = node g0 + k; 0xFFFF = invalid)
    typedef u32 Tag;         // id16 | rep16 << 16
    u32* ir; float2* b0; float2* b1; float2* b2;     // {lx, ly} {lz, hx} {hy, hz}: 8-byte LDS accesses (2 LDS cycles per wave-instruction; a 4-byte one takes 4)
    u32 g0, ni;
    static __device__ __forceinline__ Tag invalid_tag() { return 0xFFFFFFFFu; }
    static __device__ __forceinline__ bool is_valid(Tag t) { return (t & 0xFFFFu) != 0xFFFFu; }
    __device__ __forceinline__ u32 id(Tag t) const { const u32 ie = t & 0xFFFFu; return g0 + ie + ((0u - (ie >> 15)) & (ni - 0x8000u)); }   // (valid tags only)
    __device__ __forceinline__ u32 rep(Tag t) const { return g0 + (t >> 16); }
    __device__ __forceinline__ Tag with_node(Tag t, u32 node) const { return (t & 0xFFFF0000u) | (node - g0); }
    __device__ __forceinline__ void load(u32 pos, Tag& t, Box& b) const {
        t = ir[pos];
        const float2 q0 = b0[pos], q1 = b1[pos], q2 = b2[pos];
        b = { q0.x, q0.y, q1.x, q1.y, q2.x, q2.y };
    }
    __device__ __forceinline__ void store(u32 pos, Tag t, const Box& b) const {
        ir[pos] = t; b0[pos] = make_float2(b.lx, b.ly); b1[pos] = make_float2(b.lz, b.hx); b2[pos] = make_float2(b.hy, b.hz);
    }
    __device__ __forceinline__ Box load_box(u32 pos) const {
        const float2 q0 = b0[pos], q1 = b1[pos], q2 = b2[pos];
        return { q0.x, q0.y, q1.x, q1.y, q2.x, q2.y };
    }
    __device__ __forceinline__ Tag tag_at(u32 pos) const { return ir[pos]; }
    __device__ __forceinline__ void invalidate(u32 pos) const { ir[pos] = 0xFFFFFFFFu; }
};
struct WaveList {            // k_hploc_ext: a wave's two 32-slot work lists with full 32-bit ids / reps (8-byte tag + three 8-byte box planes per position)
    typedef u64 Tag;         // id | rep << 32
    u64* ir; float2* b0; float2* b1; float2* b2;
    static __device__ __forceinline__ Tag invalid_tag() { return ~0ull; }
    static __device__ __forceinline__ Tag make(u32 id, u32 rep) { return (u64)id | ((u64)rep << 32); }
    static __device__ __forceinline__ bool is_valid(Tag t) { return (u32)t != INV; }
    __device__ __forceinline__ u32 id(Tag t) const { return (u32)t; }
    __device__ __forceinline__ u32 rep(Tag t) const { return (u32)(t >> 32); }
    __device__ __forceinline__ Tag with_node(Tag t, u32 node) const { return (t & 0xFFFFFFFF00000000ull) | (u64)node; }
    __device__ __forceinline__ void load(u32 pos, Tag& t, Box& b) const {
        t = ir[pos];
        const float2 q0 = b0[pos], q1 = b1[pos], q2 = b2[pos];
        b = { q0.x, q0.y, q1.x, q1.y, q2.x, q2.y };
    }
    __device__ __forceinline__ void store(u32 pos, Tag t, const Box& b) const {
        ir[pos] = t; b0[pos] = make_float2(b.lx, b.ly); b1[pos] = make_float2(b.lz, b.hx); b2[pos] = make_float2(b.hy, b.hz);
    }
    __device__ __forceinline__ Box load_box(u32 pos) const {
        const float2 q0 = b0[pos], q1 = b1[pos], q2 = b2[pos];
        return { q0.x, q0.y, q1.x, q1.y, q2.x, q2.y };
    }
    __device__ __forceinline__ Tag tag_at(u32 pos) const { return ir[pos]; }
    __device__ __forceinline__ void invalidate(u32 pos) const { ir[pos] = ~0ull; }
};
// One task per 32-lane half.  In: have / final_ (uniform per half), cnt, and the lane's cluster (tag, b; invalid beyond cnt) as loaded from the list.
// Out: cnt survivors, the lane's cluster of slot `slot`, and the list holding them at base + [0, cnt).  lim: highest valid list position (clamp).
// ZERO_WT (tile kernel of the overlapped schedule): the node stores that involve node 0 — node 0 itself and the node that points at it — are write-through although
// AGENT is off: the root task of k_hploc_live, which runs while this launch's plain stores may still sit in an XCD's L2, reads node 0 and rewrites that link.
// IL: interleaved lane layout (nn_search_il): slot is NOT lane & 31; below_il = the half's lanes that hold lower slots.
template <bool AGENT, typename List, bool WIDE = false, bool ZERO_WT = false, bool IL = false>
__device__ __forceinline__ void ploc_rounds_lds(bool have, bool final_, u32& cnt_io, typename List::Tag& tag_io, Box& b_io, u32 base, u32 nl, u32 rbase, u32 lim,
                                                const List& list, bvh2_node* nodes, u32* zero_parent, int lane, int slot, int hbase, u64* nn,
                                                u32* rclk = nullptr, u32 below_il = 0u) {
    // rclk (measurement build, ABL_ROUND_CLOCK): wave-uniform sums {wave-rounds, shader-clock ticks spent in them, active halves} — tools/round_clock.py
    typename List::Tag tag = tag_io;
    u32 cnt = cnt_io;
    Box b = b_io;
    const u32 below = IL ? below_il : (1u << slot) - 1u;  // the half's lanes that hold lower slots
    const u32 threshold = final_ ? 1u : HP_HALF;
    // a task that needs no round only left-packs its right child's clusters (done up front: a helping half's b does not survive the loop, HPB_WIDE)
    if (have && cnt <= threshold && (u32)slot >= nl && (u32)slot < cnt) list.store(base + (u32)slot, tag, b);
    while (__ballot(have && cnt > threshold)) {
#ifdef ABL_ROUND_CLOCK
        const u64 rc_t0 = __builtin_amdgcn_s_memtime();
#endif
        const bool act = have && cnt > threshold;
        u32 nbr;
        bool wide = false;
        u64 am = 0ull;
        if (WIDE) { am = __ballot(act); wide = ((u32)am == 0u) != ((u32)(am >> 32) == 0u); }     // exactly one half of the wave runs a task (wave-uniform)
        if (WIDE && wide) {
            // nn_search_wide: the other half helps.  The task's parameters come from its half's first lane; every lane (re)reads the cluster of its slot from the task's
            // list (the task's own lanes hold exactly that already) and the one its chain starts from.  b_io of a helping half is clobbered: the tile kernel does not use it.
            const int ah = (u32)am == 0u ? 32 : 0;
            const u32 a_cnt = (u32)__builtin_amdgcn_readlane((int)cnt, ah), a_base = (u32)__builtin_amdgcn_readlane((int)base, ah);
            const u32 a_nl = (u32)__builtin_amdgcn_readlane((int)nl, ah), a_rbase = (u32)__builtin_amdgcn_readlane((int)rbase, ah);
            const u32 off = hbase == ah ? 0u : (u32)(HP_RADIUS / 2);
            const u32 t1 = (u32)slot + off;
            const u32 p0 = (u32)slot < a_nl ? a_base + (u32)slot : a_rbase + (u32)slot, p1 = t1 < a_nl ? a_base + t1 : a_rbase + t1;
            b = list.load_box(p0 < lim ? p0 : lim);
            const Box st = list.load_box(p1 < lim ? p1 : lim);
            nbr = (u32)nn_search_wide(b, st, off, a_cnt, ah + slot, slot, lane, nn) & 31u;
        } else if (IL) {
            const u32 p1 = (u32)slot + 1u < nl ? base + (u32)slot + 1u : rbase + (u32)slot + 1u;
            nbr = (u32)nn_search_il(b, list, p1 < lim ? p1 : lim, act, cnt, slot, nn + hbase) & 31u;
        } else {
            const u32 raw = (u32)nn_search<true>(b, act, cnt, lane, slot, nn);
            nbr = raw & 31u;
#if defined(ABL_EXTRA_VALU) || defined(ABL_EXTRA_BPERM)     // (keeps nn_search's in-situ probes alive in the tile kernel: the mask above would let the compiler drop them)
            if (raw & 64u) cnt = 0u;
#endif
        }
        // mergeClusters (:126-190): the neighbour's choice (low word of its key) and its record, read in one go
        const bool in = act && (u32)slot < cnt;
        const u32 pn = nbr < nl ? base + nbr : rbase + nbr;
#if HP_NN_LDS == 1 || HP_NN_LDS == 3
        const u32 nbr_of_nbr = (u32)nn[hbase + (int)nbr];              // (the key word's low half IS the neighbour's choice)
#else
        const u32 nbr_of_nbr = (u32)__shfl((int)nbr, hbase + (int)nbr);
#endif
        typename List::Tag tag_nb; Box bn;
        list.load(pn < lim ? pn : lim, tag_nb, bn);
        // the neighbour's key and record come back in ONE LDS round trip: without this the compiler sinks the box reads into the merge branch below,
        // behind the wait for the key (a third dependent round trip per round)
        asm volatile("" : "+v"(bn.lx), "+v"(bn.ly), "+v"(bn.lz), "+v"(bn.hx), "+v"(bn.hy), "+v"(bn.hz));
        const bool mutual = in && nbr_of_nbr == (u32)slot;
        const bool merge = mutual && (u32)slot < nbr;
        const bool absorbed = mutual && (u32)slot > nbr;
        // (the union, the node index and the new tag are computed by every lane and selected: loads that only feed a divergent branch are sunk into
        // it by the compiler and then wait for the neighbour's key first — one more dependent LDS round trip per round)
        const Box bu = box_union(b, bn);
        u32 at = list.rep(tag_nb) - 1u;                      // the absorbed partner's rep is retired here, once
        u32 l = list.id(tag), r = list.id(tag_nb);
        if (merge) {
            if (final_ && cnt == 2u && at != 0u) {
                // the root must be node 0 (:165-167 makes the last allocation 0): move node 0's occupant to the root's natural slot and re-point its parent
                const u64* q0 = reinterpret_cast<const u64*>(nodes);
                u64* qs = reinterpret_cast<u64*>(nodes + at);
                const u64 w0 = ld_agent(q0 + 0), w1 = ld_agent(q0 + 1), w2 = ld_agent(q0 + 2), w3 = ld_agent(q0 + 3);
                st_agent(qs + 0, w0); st_agent(qs + 1, w1); st_agent(qs + 2, w2); st_agent(qs + 3, w3);
                if (l == 0u) l = at;
                else if (r == 0u) r = at;
                else {
                    const u32 pw = ld_agent(zero_parent);
                    st_agent(reinterpret_cast<u32*>(nodes + (pw >> 1)) + (pw & 1u), at);
                }
                at = 0u;
            } else if (l == 0u || r == 0u) st_agent(zero_parent, (at << 1) | (r == 0u ? 1u : 0u));   // who points at node 0
            if (AGENT || (ZERO_WT && (at == 0u || l == 0u || r == 0u))) node_store_agent(nodes + at, l, r, bu); else node_store_plain(nodes + at, l, r, bu);
        }
        b.lx = merge ? bu.lx : b.lx; b.ly = merge ? bu.ly : b.ly; b.lz = merge ? bu.lz : b.lz;
        b.hx = merge ? bu.hx : b.hx; b.hy = merge ? bu.hy : b.hy; b.hz = merge ? bu.hz : b.hz;
        tag = merge ? list.with_node(tag, at) : tag;
        // compaction (:176-187): survivors and merged clusters write their record to their rank — in place: every read of this round has been issued
        // (a wave's LDS operations execute in order) — and every lane reads the record of its slot back
        const bool keep = in && !absorbed;
        const u32 kh = (u32)(__ballot(keep) >> hbase);
        const u32 newcnt = (u32)__popc(kh);
        if (keep) list.store(base + (u32)__popc(kh & below), tag, b);
        typename List::Tag t2; Box b2;
        list.load(base + (u32)slot < lim ? base + (u32)slot : lim, t2, b2);
        if (act) {
            cnt = newcnt; nl = 32u;
            b = b2; tag = (u32)slot < newcnt ? t2 : List::invalid_tag();
        }
#ifdef ABL_EXTRA_TRIP    // in-situ probe: one more DEPENDENT LDS round trip per round (a 4-byte read whose address depends on the read-back, result waited for)
        { u32 x = (u32)nn[(__float_as_uint(b.lx) >> 29) + (u32)(lane & 56)];
          asm volatile("" : "+v"(x));
          if (x == 0x12345u) cnt = 0u; }
#endif
#ifdef ABL_ROUND_CLOCK
        if (rclk) {
            asm volatile("" : "+v"(b.lx), "+v"(tag));                     // (the round's read-back has landed when the clock is read)
            const u64 am2 = __ballot(act);
            rclk[0] += 1u; rclk[1] += (u32)(__builtin_amdgcn_s_memtime() - rc_t0); rclk[2] += ((u32)am2 != 0u ? 1u : 0u) + ((u32)(am2 >> 32) != 0u ? 1u : 0u);
        }
#endif
    }
    tag_io = tag; cnt_io = cnt; b_io = b;
}

// ---- the asynchronous part: run ready merge tasks, two per pass (one per 32-lane half of the wave), then hand the finished
// range to the parent node (dep_arrive); a lane that completes its parent runs it next.  No waiting anywhere.
// one pass: the first two ready lanes' tasks run (rm = ballot(ready), non-zero); their owners move on to the parent if they complete it
template <bool SETUP, typename K>
__device__ __forceinline__ void climb_pass(u64 rm, bool& ready, u32& pc, u32& L, u32& R, const bvh_aabb* __restrict__ boxes, const K* __restrict__ skeys,
                                           const u32* __restrict__ svals, bvh_primref* leaves, bvh2_node* nodes, bvh2_node* recs,
                                           u64* dep, u32* zero_parent, u32 ni, int lane, u64* nn) {
    const int half = lane >> 5, slot = lane & 31, hbase = half << 5;
    {
        const int ownA = __ffsll((unsigned long long)rm) - 1;
        const u64 rm2 = rm & (rm - 1);
        const int ownB = rm2 ? __ffsll((unsigned long long)rm2) - 1 : -1;
        const int own = half ? ownB : ownA;
        const bool have = own >= 0;
        const int osrc = have ? own : 0;
        const u32 tL = (u32)__shfl((int)L, osrc), tR = (u32)__shfl((int)R, osrc), tP = (u32)__shfl((int)pc, osrc);
        // the owners look up their parent now: the four key loads fly while the task runs
        const bool owner = ready && (lane == ownA || lane == ownB);
        u32 q = INV;
        if (owner && !(L == 0u && R == ni)) q = parent_gap(L, R, ni, [&](u32 a, u32 b) { return closer(skeys, a, b); });

        HpWork w = load_work<SETUP>(have, tL, tR, tP, boxes, svals, leaves, recs, ni, slot, hbase);
        ploc_rounds(w, nodes, zero_parent, ni, lane, slot, hbase, nn);
        // storeIndices (:208-218): the <= 16 survivors of a non-root range, INVALID-terminated
        const bool left_child = __shfl((int)(q == R), osrc) != 0;      // (the owner's q: this range is its parent's left child)
        if (have && !w.final_ && slot < 16) node_store_agent(recs + rec_base(left_child, tL, tR) + slot, w.id, w.rep, w.b);

        if (owner) {
            ready = false;
            if (q != INV) {
                drain_stores();                 // the wave's node / record stores are in memory before the count moves
                ready = dep_arrive(dep, q, q == R ? dep_word(1u, L, 0u) : dep_word(1u, 0u, R), L, R);
                pc = q;
            }
        }
    }
}
template <bool SETUP, typename K>
__device__ __forceinline__ void async_climb(bool ready, u32 pc, u32 L, u32 R, const bvh_aabb* __restrict__ boxes, const K* __restrict__ skeys,
                                            const u32* __restrict__ svals, bvh_primref* leaves, bvh2_node* nodes, bvh2_node* recs,
                                            u64* dep, u32* zero_parent, u32 ni, int lane, u64* nn) {
    while (true) {
        const u64 rm = __ballot(ready);
        if (!rm) break;
        climb_pass<SETUP>(rm, ready, pc, L, R, boxes, skeys, svals, leaves, nodes, recs, dep, zero_parent, ni, lane, nn);
    }
}

// Small inputs: one launch.  Phase 1: range of every gap from the keys; big nodes enter the dependency protocol.  Phase 2: climb.
#ifndef HPA_OCC
#define HPA_OCC 7        // waves per SIMD of the one-launch kernel (68 VGPRs)
#endif
template <typename K>
__global__ __launch_bounds__(HP_BLOCK, HPA_OCC) void k_hploc(const bvh_aabb* __restrict__ boxes, const K* __restrict__ skeys,
                                                    const u32* __restrict__ svals, bvh_primref* leaves,
                                                    bvh2_node* nodes, bvh2_node* recs, u64* dep, u32* zero_parent, u32 n, int dbg) {
    const int lane = tid_x() & (WAVE - 1);
    const u32 ni = n - 1;
    u32 pc = bid_x() * HP_BLOCK + tid_x();      // LBVH gap / node this lane currently speaks for
    u32 L = 0, R = 0;
    bool ready = false;

    // The block's key window [g0 - 256, g0 + 512] sits in LDS: almost every probe of the common-prefix searches lands there
    // (a dependent L2 round trip per probe otherwise); only ranges reaching beyond the window probe global memory.
    __shared__ K s_keys[HP_BLOCK * 3 + 1];
    const int g0 = (int)(bid_x() * HP_BLOCK);
    const int w0 = g0 - HP_BLOCK;
    for (int k = tid_x(); k < HP_BLOCK * 3 + 1; k += HP_BLOCK) { const int j = w0 + k; s_keys[k] = (j >= 0 && j < (int)n) ? skeys[j] : (K)0; }
    __syncthreads();
    auto key_at = [&](int j) -> K { return ((u32)(j - w0) <= (u32)(HP_BLOCK * 3)) ? s_keys[j - w0] : skeys[j]; };
    if (pc < ni) {
        const int p = (int)pc;
        const K kp = key_at(p);
        const int c0 = plen(kp, (u32)p, key_at(p + 1), (u32)p + 1u);             // common prefix length of the node
        auto inside = [&](int j) -> bool { return j >= 0 && j < (int)n && shares_prefix(key_at(j), (u32)j, kp, (u32)p, c0); };
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
            const bool lbig = (pc - L + 1) > HP_HALF, rbig = (R - pc) > HP_HALF;
            const u32 e = (lbig ? 1u : 0u) + (rbig ? 1u : 0u);
            if (e == 0) ready = true;
            else ready = dep_arrive(dep, pc, dep_word(3u - e, lbig ? 0u : L, rbig ? 0u : R), L, R);
        }
    }
    if (dbg == 1) return;
    __shared__ u64 s_nn[HP_BLOCK / WAVE][WAVE];
    async_climb<true>(ready, pc, L, R, boxes, skeys, svals, leaves, nodes, recs, dep, zero_parent, ni, lane, s_nn[tid_x() / WAVE]);
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
// Nodes whose range crosses a block boundary ("external": the ancestors of the T-aligned gaps) use the dependency protocol:
// the block publishes the records of its maximal local ranges and its external nodes' own contributions; nodes completed by
// that are queued (HPQ_SUB sub-queues, one atomic per block) for k_hploc_ext, which climbs from there.  Nothing this kernel
// writes is read before the next launch, so all its stores are plain cached stores and nothing is drained.
// =====================================================================================================================
constexpr u32 HPQ_SUB = 64;        // sub-queues (a single queue head would serialise one atomic per block)
constexpr u32 HPQ_LOCAL = 16;      // ready items a block aggregates in LDS before falling back to one atomic per item (typically 3-6; LDS: 7 blocks per CU need <= 23040 B each)

// LIVE (overlapped schedule: k_hploc_live consumes the queue while the tile kernel fills it): a slot is its own flag — both words are zero before the build, each is written
// once (pc + 1 and {L, R}: R >= 1, so neither is zero) by a write-through store, and the consumer that holds the slot's ticket takes the item when it has seen BOTH non-zero
// and zeroes them again.  No order between the two stores is needed.
template <bool LIVE = false>
__device__ __forceinline__ void queue_put(u32* q_pc, u64* q_rng, u32 q_cap, u32 sub, u32 at_in_sub, u32 pc, u32 L, u32 R) {
    const size_t at = (size_t)sub * q_cap + at_in_sub;
    if (LIVE) { if (at_in_sub < q_cap) { st_agent(q_pc + at, pc + 1u); st_agent(q_rng + at, (u64)L | ((u64)R << 32)); } }
    else { q_pc[at] = pc; q_rng[at] = (u64)L | ((u64)R << 32); }
}
constexpr u32 HPQ_HEAD_WORDS = HPQ_SUB * 32u + 32u;  // words of queue_count a build clears: the padded heads + the line of HPQ_DONE_WORD
constexpr u32 HPQ_DONE_WORD = HPQ_SUB * 32u;     // queue_count[HPQ_DONE_WORD]: tiles of the overlapped schedule that have handed over (cleared with the heads)

template <typename K, int T, int NT, int OCC, bool LIVE = false>
__global__ __launch_bounds__(NT, OCC) void k_hploc_block(const bvh_aabb* __restrict__ boxes, const K* __restrict__ skeys,
                                                    const u32* __restrict__ svals, bvh_primref* __restrict__ leaves,
                                                    bvh2_node* nodes, bvh2_node* recs, u64* dep, u32* zero_parent,
                                                    u32* q_pc, u64* q_rng, u32* q_count, u32 q_cap, u32 n, int dbg, const float4* __restrict__ tris) {
    constexpr int PER = T / NT;                      // leaf positions (and gaps) per thread
    constexpr int NW = NT / WAVE;
    static_assert(T % NT == 0 && T <= 16384, "block-local HPLOC tile");
    constexpr int NLEV = KeyBits<K>::value;          // hierarchy levels = bits of the augmented key (64 / 96)
    constexpr int KM = 18;                           // key margin: the hand-over probes up to 17 leaves beyond the tile's rims (small children of external
#ifndef HPB_IL
#define HPB_IL 1         // 1 (default since round 6): interleaved lane layout of the tile kernel's rounds (nn_search_il: the row shifts become DPP operands of v_min / v_max,
                         //    no branch around a candidate: -45 VALU instructions and -8 scalar branches per round).  Round 3 measured it at nothing with both ends of a pair
                         //    minimised by LDS atomics; with the own end in a register pair (HP_NN_LDS = 3) same box, three alternating pairs: 10 M 0.5690 -> 0.5600 ms,
                         //    2 M 0.1389 -> 0.1367 (LEADS.md row 94).  0: the wave_shl chain of nn_search (the alt variant; HPB_WIDE needs it)
#endif
#ifndef HPB_WIDE
#define HPB_WIDE 0       // A/B switch (off: measured, no gain — LEADS.md row 64).  1: a round in which only one half of the wave still has a task runs that task on the whole wave (nn_search_wide: four candidates per lane)
#endif
                         //    needed 70: spills in the rounds); since the rounds no longer keep a task's box and tag alive for a store behind the loop (ploc_rounds_lds: the
                         //    left-pack of a task that needs no round runs up front; 70 -> 66 VGPRs, tile kernel 0.644 -> 0.633 ms by itself) the kernel fits 64 registers with
                         //    two spilled: 10 M 0.6396 -> 0.6255 ms, 2 M 0.158 -> 0.1535 (production flags, three runs each; HPB_OCC = 8 goes with it)
    constexpr int NLV = NLEV;
    constexpr size_t KN_BYTES = sizeof(K) * (T + 2 * KM) > sizeof(u64) * (NT + 8) ? sizeof(K) * (T + 2 * KM) : sizeof(u64) * (NT + 8);
    __shared__ u64 s_kn[(KN_BYTES + 7) / 8];
    K* const s_key = reinterpret_cast<K*>(s_kn);
    u64 (* const s_nn)[WAVE] = reinterpret_cast<u64 (*)[WAVE]>(s_kn);
    // work lists: per position the cluster's id and rep, tile-relative in 16 bits (a cluster merged inside the tile absorbs a
    // partner whose first leaf lies in the tile, so node index = rep' - 1 is tile-local too), and its box (SoA)
    __shared__ u32 e_ir[T];                          // id | rep << 16, tile-relative (TileList)
    __shared__ float2 e_bx[3 * T + 2];               // {lx, ly} | {lz, hx} | {hy, hz} per position; the three planes (T + 1) * 8 bytes apart so that the compiler cannot
                                                     // fuse two plane accesses into one ds_read2(st64)_b64 (8 LDS cycles; two ds_read_b64 take 2 each)
    float2* const e_b0 = e_bx; float2* const e_b1 = e_bx + T + 1; float2* const e_b2 = e_bx + 2 * T + 2;
    __shared__ u32 m_range[T];                       // per gap (relative): L | parent side << 15 | R << 16 of a local big node; 0xFFFF in the high half: the range leaves the block
    __shared__ unsigned short s_task[T];             // local big nodes grouped by level; later: the maximal local nodes to publish
    __shared__ u32 s_cnt[NLV], s_off[NLV];
    __shared__ u64 s_lvmask[2];                      // the non-empty levels (bit lv of word lv / 64): the level loop visits only those
    __shared__ u32 s_npub, s_nready, s_qbase, s_ncand;
    __shared__ u32 r_pc[HPQ_LOCAL], r_L[HPQ_LOCAL], r_R[HPQ_LOCAL];
#ifndef HPB_LIVE_PAD
#define HPB_LIVE_PAD 0   // overlapped schedule: extra LDS bytes per tile, so that fewer tiles fill a CU and k_hploc_live's workgroup finds room beside them
#endif
    __shared__ u32 s_livepad[LIVE && HPB_LIVE_PAD > 0 ? HPB_LIVE_PAD / 4 : 1];
    if (LIVE && HPB_LIVE_PAD > 0 && n == 0xFFFFFFFFu) s_livepad[tid_x() % (HPB_LIVE_PAD > 0 ? HPB_LIVE_PAD / 4 : 1)] = 1u;      // (never true: keeps the allocation)
#ifdef ABL_LDS_PAD       // in-situ probe: fewer workgroups per CU (is the kernel bound by latency x occupancy?)
    __shared__ u32 s_pad[ABL_LDS_PAD / 4];
    if (tid_x() == 0 && n == 0xFFFFFFFFu) s_pad[bid_x() % (ABL_LDS_PAD / 4)] = 1u;
#endif

    const int tid = tid_x(), lane = tid & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave: an SGPR — the level loop's task dealing is scalar arithmetic and scalar branches)
    const int half = lane >> 5, slot = lane & 31, hbase = half << 5;
#ifdef ABL_TILE_PHASES    // measurement build: where a tile spends its life, as thread 0 sees it (clock ticks summed over the tiles into words 4.. of every sub-queue's padded head;
    __shared__ u64 s_ph[8];                                               // (stamps go straight to LDS: no register lives across the kernel for them)          // tools/tile_phases.py): start, staged, ranges + level sort done, level loop done, hand-over done; barrier waits of the loop
#define TILE_PHASE(K) do { if (tid_x() == 0) s_ph[K] = __builtin_amdgcn_s_memtime(); } while (0)   /* the CU's own shader clock: the chip-wide s_memrealtime serialises (17 stamps per tile tripled the kernel) */
#else
#define TILE_PHASE(K) do { } while (0)
#endif
    TILE_PHASE(0);
#ifdef ABL_TILE_PHASES
    if (tid_x() == 0) { s_ph[6] = 0ull; s_ph[7] = 0ull; }
#endif
    const u32 ni = n - 1;
    const u32 g0 = bid_x() * (u32)T;
    const u32 nleaf = (n - g0) < (u32)T ? (n - g0) : (u32)T;
    const u32 sub = bid_x() % HPQ_SUB;
    const TileList tl{ e_ir, e_b0, e_b1, e_b2, g0, ni };

    // ---- stage the block: leaves (SetupClusters :44-47, fused), keys -------------------------------------------------
    // Every load of a dependency level is issued before the first one is waited for: the key window and the PER primitive indices together, then the PER box gathers.
    // (Indices are clamped, not branched around: with a branch per leaf the compiler emitted index -> wait -> box -> wait per leaf and then the key window one load
    // at a time — seven dependent HBM round trips at the start of every tile where two suffice; round 4, found in the ISA.)
    // The PER box gathers of a thread are the exception (HPB_STAGE_SERIAL_GATHER): both in flight at once measured SLOWER at 10 M, where the gather misses every cache
    // (tile kernel, production flags, ms at 10 M / 2 M: before 0.6115 / 0.149; everything together 0.6235 / 0.144; gathers one after the other 0.595 / 0.143).
    constexpr int KW = (T + 2 * KM + NT - 1) / NT;
    K kwin[KW];
#pragma unroll
    for (int q = 0; q < KW; ++q) {
        const int k = tid + q * NT;
        const long long j = (long long)g0 - KM + k;
        const bool in = k < T + 2 * KM && j >= 0 && j < (long long)n;
        kwin[q] = skeys[in ? j : (long long)g0];
        if (!in) kwin[q] = (K)0;
    }
    u32 prim_[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) { const u32 k = (u32)tid + (u32)i * NT; prim_[i] = svals[g0 + (k < nleaf ? k : nleaf - 1u)]; }
    Box box_[PER];
    if (tris) {
#pragma unroll
        for (int i = 0; i < PER; ++i) box_[i] = tri_box_gather(tris + (size_t)prim_[i] * 4);
    } else {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            box_[i] = box_gather(boxes + prim_[i]);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(box_[i].lx), "+v"(box_[i].ly), "+v"(box_[i].lz), "+v"(box_[i].hx), "+v"(box_[i].hy), "+v"(box_[i].hz) :: "memory");
        }
    }
#pragma unroll
    for (int q = 0; q < KW; ++q) { const int k = tid + q * NT; if (k < T + 2 * KM) s_key[k] = kwin[q]; }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const u32 k = (u32)tid + (u32)i * NT;
        if (k < nleaf) {
            const u32 g = g0 + k, prim = prim_[i];
            const Box b = box_[i];
            float* f = reinterpret_cast<float*>(leaves + g);
            reinterpret_cast<u32*>(f)[0] = prim;
            f[1] = b.lx; f[2] = b.ly; f[3] = b.lz; f[4] = b.hx; f[5] = b.hy; f[6] = b.hz;
            e_ir[k] = 0x8000u | k | (k << 16);
            e_b0[k] = make_float2(b.lx, b.ly); e_b1[k] = make_float2(b.lz, b.hx); e_b2[k] = make_float2(b.hy, b.hz);
        }
    }
    if (tid < NLV) s_cnt[tid] = 0u;
    if (tid == 0) { s_npub = 0u; s_nready = 0u; s_ncand = 0u; }
    __syncthreads();
    TILE_PHASE(1);
    if (dbg == 1) return;

    // ---- ranges of the block's gaps, clamped to the window [g0-1, g0+T]; a range touching the window's rim is external
    constexpr u32 M_EXT_TAG = 0xFFFF0000u;          // m_range of an external gap: tag | its contribution (see below)
    auto m_is_ext = [](u32 w) -> bool { return (w >> 16) == 0xFFFFu; };
    const int jmin = g0 ? (int)g0 - 1 : 0;
    const int jmax = (g0 + (u32)T <= ni) ? (int)(g0 + (u32)T) : (int)ni;
    auto wkey = [&](int j) -> K { return s_key[j - (int)g0 + KM]; };
    int my_lv[PER]; u32 my_pos[PER], my_gap[PER];
    u32 my_cls[PER];
    auto lv_total = [](u32 w) -> u32 { return (w & 63u) + ((w >> 6) & 63u) + ((w >> 12) & 63u) + ((w >> 18) & 63u) + ((w >> 24) & 63u); };
    // Only ~9 % of the gaps are merge tasks, another few per cent are lopsided small nodes or sit near a rim of the tile, yet the two binary searches below cost
    // ~160 VALU instructions per gap: two probes at p - 8 and p + 9 first — neither inside means the node spans at most [p - 7, p + 8], 16 leaves, inside the
    // tile: no task, not external — and only the gaps that survive are compacted (s_task is free until the level sort) and searched, one per thread.
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const u32 k = (u32)tid + (u32)i * NT;
        const u32 pc = g0 + k;
        if (k < nleaf && pc < ni) {
            const int p = (int)pc;
            const K kp = wkey(p);
            const int c0 = plen(kp, (u32)p, wkey(p + 1), (u32)p + 1u);
            const bool small = p - 8 >= jmin && p + 9 <= jmax && !shares_prefix(wkey(p - 8), (u32)(p - 8), kp, (u32)p, c0) && !shares_prefix(wkey(p + 9), (u32)(p + 9), kp, (u32)p, c0);
            if (small) m_range[k] = 0u;
            else s_task[atomicAdd(&s_ncand, 1u)] = (unsigned short)k;
        }
    }
    __syncthreads();
    const u32 ncand = s_ncand;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const u32 ci = (u32)tid + (u32)i * NT;
        const bool on = ci < ncand;
        const u32 k = on ? (u32)s_task[ci] : 0u;
        const u32 pc = g0 + k;
        my_lv[i] = -1; my_pos[i] = 0; my_gap[i] = k;
        if (on) {
            const int p = (int)pc;
            const K kp = wkey(p);
            const int c0 = plen(kp, (u32)p, wkey(p + 1), (u32)p + 1u);
            auto inside = [&](int j) -> bool { return shares_prefix(wkey(j), (u32)j, kp, (u32)p, c0); };
            // the positions sharing the node's prefix are contiguous around p: plain binary searches over the window for the two
            // ends (a fixed ~log2(T) probes per side; an exponential search costs the wave its longest lane: ~2x as many)
            int lo = p, hi = p + 1;
            {   int a = jmin, b = p;                                             // first inside position in [jmin, p]
                while (a < b) { const int mid = (a + b) >> 1; if (inside(mid)) b = mid; else a = mid + 1; }
                lo = a; }
            {   int a = p + 1, b = jmax;                                         // last inside position in [p+1, jmax]
                while (a < b) { const int mid = (a + b + 1) >> 1; if (inside(mid)) a = mid; else b = mid - 1; }
                hi = a; }
#ifdef HPB_CUT            // measurement build (LEADS.md rows 15, 84): local nodes of more than HPB_CUT leaves are handed to k_hploc_ext like the nodes that cross the tile
            const bool ext = lo < (int)g0 || hi > (int)(g0 + (u32)T - 1u) || (hi - lo + 1) > (int)(HPB_CUT);
#else
            const bool ext = lo < (int)g0 || hi > (int)(g0 + (u32)T - 1u);
#endif
            m_range[k] = 0u;
            if (ext) {
                // An external node's own contribution to its dependency word (the hand-over adds it): which children are big — child [L, p] iff leaf p - 16 shares
                // the prefix, child [p + 1, R] iff leaf p + 17 does — and the far ends of the small ones.  Computed HERE, from the key window (it reaches 18 positions
                // beyond the tile): the hand-over then needs no key at all, and the lean layout no second read of the window (round 4).
                auto inside_n = [&](int j) -> bool { return j >= 0 && j < (int)n && inside(j); };
                const bool lbig = inside_n(p - (int)HP_HALF), rbig = inside_n(p + 1 + (int)HP_HALF);
                int elo = p, ehi = p + 1;
                if (!lbig) { for (int t = 8; t > 0; t >>= 1) if (inside_n(elo - t)) elo -= t; }          // L in [p-15, p]
                if (!rbig) { for (int t = 8; t > 0; t >>= 1) if (inside_n(ehi + t)) ehi += t; }          // R in [p+1, p+16]
                m_range[k] = M_EXT_TAG | (lbig ? 1u : 0u) | (rbig ? 2u : 0u) | ((u32)(p - elo) << 2) | ((u32)(ehi - (p + 1)) << 6);
            }
            if (!ext && (u32)(hi - lo + 1) > HP_HALF) {
                // (bit 15: the node's parent is the gap on its LEFT, lo - 1 — findParent (:66-81) on the window's keys; the hand-over publishes the node if that parent is external)
                const u32 gl = (u32)lo, gr = (u32)hi;
                const u32 pq = parent_gap(gl, gr, ni, [&](u32 a, u32 b2) { return plen(wkey((int)a), a, wkey((int)a + 1), a + 1u) > plen(wkey((int)b2), b2, wkey((int)b2 + 1), b2 + 1u); });
                m_range[k] = (u32)(lo - (int)g0) | (pq == gr ? 0u : 0x8000u) | ((u32)(hi - (int)g0) << 16);
                my_lv[i] = NLEV - 1 - c0;
                // the level's tasks are grouped by size class (6-bit counters packed in the level's word: at most T / 17 <= 60 disjoint tasks per level), so that
                // the two tasks of a wave pass need about the same number of rounds (a pass lasts as long as the longer of its two tasks)
                const u32 sz = (u32)(hi - lo + 1);
                my_cls[i] = sz > 32u ? 4u : (sz - 17u) >> 2;
                my_pos[i] = (atomicAdd(&s_cnt[my_lv[i]], 1u << (6u * my_cls[i])) >> (6u * my_cls[i])) & 63u;
            }
        }
    }
    __syncthreads();
    if (tid < 64) {                                  // exclusive scan of the (<= 128) level counts, two per lane
        const u32 v0 = 2 * tid < NLV ? lv_total(s_cnt[2 * tid]) : 0u, v1 = 2 * tid + 1 < NLV ? lv_total(s_cnt[2 * tid + 1]) : 0u; u32 incl = v0 + v1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = (u32)__shfl_up((int)incl, d); if (lane >= d) incl += t; }
        if (2 * tid + 1 < NLV) { s_off[2 * tid] = incl - v0 - v1; s_off[2 * tid + 1] = incl - v1; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) if (my_lv[i] >= 0) {
        const u32 w = s_cnt[my_lv[i]];
        const u32 before = lv_total(w & ((1u << (6u * my_cls[i])) - 1u));            // tasks of the lower classes on this level
        s_task[s_off[my_lv[i]] + before + my_pos[i]] = (unsigned short)my_gap[i];
    }
    __syncthreads();
    if (tid < 128) {                                                                  // (waves 0 and 1; NLV <= 128)
        const u32 tot = tid < NLV ? lv_total(s_cnt[tid]) : 0u;
        if (tid < NLV) s_cnt[tid] = tot;                                              // the level loop reads plain counts
        const u64 m = __ballot(tot != 0u);
        if (lane == 0) s_lvmask[wave] = m;
    }
    __syncthreads();
    TILE_PHASE(2);
    if (dbg == 2) return;
#ifdef BVH_ABLATION       // measurement build: merge tasks run by the tile kernel (word 2 of sub-queue 0's padded head; read through BVH_OPT_DEBUG_TASKS_LOCAL)
    if (tid == 0) { u32 t = 0; for (int lv = 0; lv < NLEV; ++lv) t += s_cnt[lv]; atomicAdd(q_count + 2, t); }
#endif

    // ---- local hierarchy, deepest level first; two tasks per wave pass ----------------------------------------------------
                         //    all seven resident tiles queue on ONE SIMD while the other three idle.  Measured (round 4): the premise is wrong — tools/probes/simd_map.hip
                         //    shows the hardware starts every workgroup's round-robin on the next SIMD (wave 0 lands on each SIMD a quarter of the time) — and the
                         //    switch changes nothing (0.6500 / 0.6515 vs 0.6511 / 0.6495 ms).  Off.
    const u32 wrot = (u32)wave;
                         //    has big children.  Waves walk their static share of the level-sorted task list in order, so whatever a task waits for sits EARLIER in some
                         //    wave's sequence: no cycle.  A wave's LDS operations execute in order: the survivors a task wrote are in LDS before its arrival count is.
    // (only the non-empty levels are visited — ~5 of 64: a scan over s_cnt cost a dependent LDS read per empty level)
#ifdef ABL_ROUND_CLOCK    // measurement build (tools/round_clock.py): how long a PLOC round takes a wave INSIDE the launch, how many of a wave's level-loop ticks are rounds
    u32 rclk[3] = { 0u, 0u, 0u };
    u32* const rclk_p = rclk;
    const u64 rc_loop0 = __builtin_amdgcn_s_memtime();
#else
    u32* const rclk_p = nullptr;
#endif
    u64 lvm[2];
    for (int i = 0; i < 2; ++i) { const u64 m = s_lvmask[i]; lvm[i] = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(m >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)m); }
    for (int lw = 0; lw < (NLEV + 63) / 64; ++lw)
    while (lvm[lw]) {
        const int lv = lw * 64 + (int)__builtin_ctzll(lvm[lw]);                        // block-uniform
        lvm[lw] &= lvm[lw] - 1ull;
        const u32 c = (u32)__builtin_amdgcn_readfirstlane((int)s_cnt[lv]);             // (block-uniform: the deal below is scalar arithmetic and a scalar loop)
        const u32 base = (u32)__builtin_amdgcn_readfirstlane((int)s_off[lv]);
        for (u32 tw = wrot * 2u; tw < c; tw += (u32)NW * 2u) {
            const u32 t = tw + (u32)half;
            const bool have = t < c;
            u32 P = 0, L = 0, R = 0;
            if (have) { P = s_task[base + t]; const u32 rg = m_range[P]; L = rg & 0x3FFFu; R = (rg >> 16) & 0x3FFFu; }
            // loadIndices (:192-206) from the LDS work lists: the first <= 16 valid entries of each child range, left-packed on the fly; the rounds
            // run on the list in place and leave the survivors at the range's first positions (storeIndices :208-218)
            const bool is_left = slot < 16;
            const u32 kk = (u32)(slot & 15);
            const u32 c_start = is_left ? L : P + 1u, c_len = is_left ? (P - L + 1u) : (R - P);
            const bool ok = have && kk < c_len && TileList::is_valid(tl.tag_at(c_start + kk));
            const u32 vb = (u32)(__ballot(ok) >> hbase);
            const u32 nl = (u32)__popc(vb & 0xFFFFu), nr = (u32)__popc(vb >> 16);
            u32 cnt = nl + nr;
            const u32 rbase = P + 1u - nl;
#if HPB_IL
            const int ts = ((lane & 15) << 1) | ((lane >> 4) & 1);                               // the task slot this lane holds (interleaved layout: nn_search_il)
            const u32 below_il = ((1u << ((lane & 15) + ((lane >> 4) & 1))) - 1u) | (((1u << (lane & 15)) - 1u) << 16);
#else
            const int ts = slot; const u32 below_il = 0u;
#endif
            const u32 sp = (u32)ts < nl ? L + (u32)ts : rbase + (u32)ts;
            TileList::Tag tag; Box b;
            tl.load(sp < (u32)T ? sp : (u32)T - 1u, tag, b);
            if (!(have && (u32)ts < cnt)) tag = TileList::invalid_tag();
            ploc_rounds_lds<false, TileList, (HPB_WIDE != 0 && HPB_IL == 0), LIVE, HPB_IL != 0>(have, false, cnt, tag, b, L, nl, rbase, (u32)T - 1u, tl, nodes, zero_parent, lane, ts, hbase, s_nn[wave], rclk_p, below_il);
            if (have && (u32)ts >= cnt && ts < 16) tl.invalidate(L + (u32)ts);          // INVALID-terminated
        }
#if defined(ABL_TILE_PHASES) && ABL_TILE_PHASES >= 2     // (2: also the waits at the levels' barriers — two more stamps per level)
        const u64 ph_b0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (tid_x() == 0) { s_ph[6] += __builtin_amdgcn_s_memtime() - ph_b0; s_ph[7] += 1ull; }
#else
        __syncthreads();
#endif
    }
#ifdef ABL_ROUND_CLOCK
    if (lane == 0) {     // per wave, into the sub-queue's padded head (words 20..24; cleared with the queue heads by every build's first kernel)
        u32* out = q_count + sub * 32u + 20u;
        atomicAdd(out + 0, rclk[0]); atomicAdd(out + 1, rclk[1]); atomicAdd(out + 2, rclk[2]);
        atomicAdd(out + 3, (u32)(__builtin_amdgcn_s_memtime() - rc_loop0)); atomicAdd(out + 4, 1u);
    }
#endif
    TILE_PHASE(3);
    if (dbg == 3) return;

    // ---- hand-over, step 1 (one thread per gap): an external node adds its own contribution (prepared with the ranges: which children are small, and
    // those children's far ends); a maximal local node (parent external) is listed for publication.  (s_task is dead as a task list: every thread is past
    // the level loop's last barrier.)  No key is read from here on.
    auto ready_push = [&](u32 pc, u32 L, u32 R) {
        const u32 at = atomicAdd(&s_nready, 1u);
        if (at < HPQ_LOCAL) { r_pc[at] = pc; r_L[at] = L; r_R[at] = R; }
        else queue_put<LIVE>(q_pc, q_rng, q_cap, sub, atomicAdd(q_count + sub * 32u, 1u), pc, L, R);   // (pathological tiles only)
    };
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const u32 k = (u32)tid + (u32)i * NT;
        const u32 pc = g0 + k;
        if (k < nleaf && pc < ni) {
            const u32 rg = m_range[k];
            if (m_is_ext(rg)) {
                const bool lbig = (rg & 1u) != 0u, rbig = (rg & 2u) != 0u;
                const u32 lo = pc - ((rg >> 2) & 15u), hi = pc + 1u + ((rg >> 6) & 15u);
                const u32 e = (lbig ? 1u : 0u) + (rbig ? 1u : 0u);
                if (e == 0u) { if (hi - lo + 1u > HP_HALF) ready_push(pc, lo, hi); }
                else {
                    u32 L, R;
                    if (dep_arrive(dep, pc, dep_word(3u - e, lbig ? 0u : lo, rbig ? 0u : hi), L, R)) ready_push(pc, L, R);
                }
            } else if (rg != 0u) {                   // local big node
                const u32 L = g0 + (rg & 0x3FFFu), R = g0 + ((rg >> 16) & 0x3FFFu);         // (bits 30..31 unused)
                const bool pleft = (rg & 0x8000u) != 0u;
                const u32 q = pleft ? L - 1u : R;
                if (q < g0 || m_is_ext(m_range[q - g0])) s_task[atomicAdd(&s_npub, 1u)] = (unsigned short)(k | (pleft ? 0x8000u : 0u));
            }
        }
    }
    __syncthreads();
    // ---- step 2 (16 lanes per node): the records of a maximal local range go to global memory, then its parent's count moves
    {
        const u32 npub = s_npub;
        const int grp = tid >> 4, sl = tid & 15;
        const u32 trips = (npub + (u32)(NT / 16) - 1u) / (u32)(NT / 16);                 // block-uniform
        for (u32 it = 0; it < trips; ++it) {
            const u32 j = it * (u32)(NT / 16) + (u32)grp;
            const bool on = j < npub;
            u32 L = 0, R = 0; bool right = false;
            if (on) {
                const u32 tk = s_task[j];
                const u32 rg = m_range[tk & 0x7FFFu];
                const u32 Lr = rg & 0x3FFFu; L = g0 + Lr; R = g0 + ((rg >> 16) & 0x3FFFu); right = (tk & 0x8000u) != 0u;
                const u32 sp = Lr + (u32)sl;
                TileList::Tag tg; Box b;
                tl.load(sp, tg, b);
                if (LIVE) node_store_agent(recs + rec_base(!right, L, R) + sl, TileList::is_valid(tg) ? tl.id(tg) : INV, tl.rep(tg), b);   // read by k_hploc_live DURING this launch
                else node_store_plain(recs + rec_base(!right, L, R) + sl, TileList::is_valid(tg) ? tl.id(tg) : INV, tl.rep(tg), b);     // read by k_hploc_ext: the kernel boundary orders it
            }
            if (LIVE) drain_stores();                // (the wave's record stores are in memory before a count moves)
            if (on && sl == 0) {                     // lane 0 of each group moves the parent's count
                const u32 q = right ? L - 1u : R;
                u32 pL, pR;
                if (dep_arrive(dep, q, right ? dep_word(1u, 0u, R) : dep_word(1u, L, 0u), pL, pR)) ready_push(q, pL, pR);
            }
        }
    }
    __syncthreads();
    TILE_PHASE(4);
    if (dbg == 4) return;
    // ---- step 3: the block's ready nodes join its sub-queue (one atomic per block)
    const u32 nready = s_nready < HPQ_LOCAL ? s_nready : HPQ_LOCAL;
    if (nready) {
        if (tid == 0) s_qbase = atomicAdd(q_count + sub * 32u, nready);
        __syncthreads();
        for (u32 i = (u32)tid; i < nready; i += (u32)NT) queue_put<LIVE>(q_pc, q_rng, q_cap, sub, s_qbase + i, r_pc[i], r_L[i], r_R[i]);
    }
    // (overlapped schedule) this tile has handed over: every reservation it made in a sub-queue's count has returned (the overflow path's before the barrier above, the
    // block's own into s_qbase), so a consumer that reads "all tiles done" and THEN a count reads the count's final value
    if (LIVE && tid == 0) __hip_atomic_fetch_add(q_count + HPQ_DONE_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef ABL_TILE_PHASES
    TILE_PHASE(5);
    if (tid == 0) {
        u32* out = q_count + sub * 32u + 4u;           // (per sub-queue head: 19 532 tiles x 8 atomics on ONE cache line took 1.3 ms)
        for (int k = 0; k < 5; ++k) atomicAdd(out + k, (u32)(s_ph[k + 1] - s_ph[k]));
        atomicAdd(out + 8, (u32)s_ph[6]); atomicAdd(out + 9, (u32)s_ph[7]); atomicAdd(out + 10, 1u);
    }
#endif
#undef TILE_PHASE
}

// One pass of k_hploc_ext.  The wave's two halves are sticky here: half h runs the task its owner lane h * 32 holds (queue items are
// dealt to lanes 0 and 32).  What differs from climb_pass is the hand-over.  A finished range normally publishes its <= 16 survivors as
// records, drains the stores, adds itself to the parent's dependency word with a returning atomic and — if that completed the parent —
// loads both children's records back: four dependent memory round trips per level, ~7 us, and the external nodes are one long chain
// (~13 levels at 10 M plus the chains along every tile boundary).  But most of the time the arriving child is the LAST contribution:
// the sibling finished long ago (a tile-local range, or a small one) — the word then already holds count 2.  So the owner first READS
// the word; if the count is 2, nobody else will touch the word again: the owner resets it, takes the parent's range from the sum, and the
// half goes on with the parent keeping the survivors IN REGISTERS as one side of the parent's work list (carry) — no record stores, no
// drain, no atomic, and only the sibling's records to load.  Otherwise the usual protocol runs unchanged.  (Whenever a node's continuation
// passes to another wave it is through the usual protocol, whose drain also covers the node stores made since.)
#ifdef ABL_EXT_TRACE     // measurement build: the tasks of more than n / 4096 leaves leave {start, end (100 MHz clock), range, hand-over taken} in the unused tail of the queue buffer
__device__ __forceinline__ void ext_trace(u64* trace, u32* trace_count, u32 cap, u64 t0, u32 L, u32 R, u32 ni, u32 kind, u32 nrounds, u64 t_loaded = 0, u64 t_rounds = 0, u64 t_done = 0) {
    if ((R - L + 1u) <= (ni + 1u) / 4096u) return;
    const u32 at = atomicAdd(trace_count, 1u);
    // e[3]: kind | rounds << 8 | (ticks from the pass's start to "work list loaded") << 16 | (... to "rounds done") << 32 (100 MHz ticks, 16 bits each)
    if (at < cap) { u64* e = trace + (size_t)at * 4u; e[0] = t0; e[1] = __builtin_amdgcn_s_memrealtime(); e[2] = (u64)L | ((u64)R << 32);
                    // (bits 48..63: ... to "hand-over done", stamped BEFORE this function's own atomic: e[1] includes the trace's round trip, this does not)
                    e[3] = (u64)kind | ((u64)nrounds << 8) | (((t_loaded - t0) & 0xFFFFull) << 16) | (((t_rounds - t0) & 0xFFFFull) << 32) | (((t_done - t0) & 0xFFFFull) << 48); }
}
#define EXT_TRACE_ARGS , u64* trace = nullptr, u32* trace_count = nullptr, u32 trace_cap = 0u
#define EXT_TRACE_PASS , trace, trace_count, trace_cap
#else
#define EXT_TRACE_ARGS
#define EXT_TRACE_PASS
#endif
struct ExtCarry { u32 id, rep; Box b; int side; };    // a half's survivors after a task (slot = lane & 31 < 16); side (owner lane): 0 none, 1 = they are
                                                      // the LEFT child of the half's next task, 2 = the RIGHT child
// LIVE (k_hploc_live, running beside the tile kernel): a small child's leaves are staged from the box array through the sorted values like the tile kernel stages them —
// both arrays are older than either launch —, not from the PrimRefs the tile kernel writes: the tile that owns a leaf may not have run yet, and its plain stores
// are not visible across XCDs before its launch ends.
template <typename K, bool LIVE = false>
__device__ __forceinline__ void ext_pass(bool& ready, u32& pc, u32& L, u32& R, ExtCarry& cw, const K* __restrict__ skeys, const bvh_aabb* __restrict__ boxes, const u32* __restrict__ svals,
                                         const bvh_primref* leaves, bvh2_node* nodes, bvh2_node* recs, u64* dep, u32* zero_parent, u32 ni, int lane, u64* nn, const WaveList& wl, u32* prof = nullptr EXT_TRACE_ARGS) {
    const int half = lane >> 5, slot = lane & 31, hbase = half << 5;
#ifdef ABL_EXT_TRACE
    const u64 tr0 = __builtin_amdgcn_s_memrealtime(); const u32 trL = L, trR = R; u32 tr_rounds = 0;
#endif
#ifdef ABL_EXT_TIMING    // measurement build: where a pass spends its cycles and which hand-over it takes (words 3.. of sub-queue 0's padded head; tools/ext_timing.py)
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const bool have = __shfl((int)ready, hbase) != 0;
    const u32 tL = (u32)__shfl((int)L, hbase), tR = (u32)__shfl((int)R, hbase), tP = (u32)__shfl((int)pc, hbase);
    const int side = __shfl(cw.side, hbase);
    const bool owner = ready && slot == 0;
    // the owner's parent: the four keys that decide it are requested now, beside the work list's loads, and compared behind them (round 4: the parent was looked up on the
    // spot, the leaves and the records were loaded one after the other — three dependent memory round trips at the start of every pass where one suffices; found in the ISA)
    const bool interior = owner && L != 0u && R != ni;
    K pk0 = (K)0, pk1 = (K)0, pk2 = (K)0, pk3 = (K)0;
    if (interior) { pk0 = skeys[R]; pk1 = skeys[R + 1u]; pk2 = skeys[L - 1u]; pk3 = skeys[L]; }

    // work list (loadIndices :192-206): one child may be the half's own survivors of the previous pass
    const bool is_left = slot < 16;
    const u32 s = (u32)(slot & 15);
    const u32 c_start = is_left ? tL : tP + 1u, c_len = is_left ? (tP - tL + 1u) : (tR - tP);
    const int csrc = (side == 2 && !is_left) ? hbase + (int)s : lane;                 // a carried RIGHT child moves from slots 0..15 to 16..31
    const u32 cid = (u32)__shfl((int)cw.id, csrc), crep = (u32)__shfl((int)cw.rep, csrc);
    const Box cb = shfl_box(cw.b, csrc);
    const bool carried = have && ((side == 1 && is_left) || (side == 2 && !is_left));
    const bool from_rec = have && !carried && c_len > HP_HALF;
    rec_v4f r0 = { 0.f, 0.f, 0.f, 0.f }, r1 = { 0.f, 0.f, 0.f, 0.f };
    if (from_rec) rec_load_agent_issue(recs + rec_base(is_left, c_start, tP) + s, r0, r1);      // (requested first: the coherent loads take longest)
    u32 id = INV, rep = INV;
    Box b = box_empty();
    if (carried) { id = cid; rep = crep; b = cb; }
    const bool leaf = have && !carried && c_len <= HP_HALF && s < c_len;
    if (leaf) { rep = c_start + s; id = ni + rep;
                if (LIVE) b = box_gather(boxes + svals[rep]);
                else b = box_load_u(reinterpret_cast<const bvh_aabb*>(reinterpret_cast<const float*>(leaves + rep) + 1)); }
    rec_wait(r0, r1);                                                                          // one wait for records, leaves and the parent's keys
    if (from_rec) { id = __float_as_uint(r0.x); rep = __float_as_uint(r0.y); b = { r0.z, r0.w, r1.x, r1.y, r1.z, r1.w }; }
    u32 q = INV;
    if (owner && !(L == 0u && R == ni)) q = L == 0u ? R : R == ni ? L - 1u : (closer_keys(pk0, pk1, R, pk2, pk3, L - 1u) ? R : L - 1u);
    const u32 vb = (u32)(__ballot(id != INV) >> hbase);
    const u32 nl = (u32)__popc(vb & 0xFFFFu), nr = (u32)__popc(vb & 0xFFFF0000u);
    HpWork w; w.cnt = nl + nr; w.tL = tL; w.have = have; w.final_ = have && tL == 0 && tR == ni;
    const int src = hbase + (((u32)slot < nl) ? slot : (int)((16 + (u32)slot - nl) & 31));
    const u32 ti = (u32)__shfl((int)id, src);
    w.rep = (u32)__shfl((int)rep, src);
    w.b = shfl_box(b, src);
    w.id = ((u32)slot < w.cnt) ? ti : INV;

#ifdef ABL_EXT_TIMING
    const u64 t1 = __builtin_amdgcn_s_memrealtime();
    u32 nrounds = 0;
    ploc_rounds(w, nodes, zero_parent, ni, lane, slot, hbase, nn, [&]() { ++nrounds; });
    const u64 t2 = __builtin_amdgcn_s_memrealtime();
#elif defined(ABL_EXT_TRACE)
    asm volatile("" : "+v"(w.b.lx), "+v"(w.id));                            // (the list is in registers before the stamp)
    const u64 tr1 = __builtin_amdgcn_s_memrealtime();
    ploc_rounds(w, nodes, zero_parent, ni, lane, slot, hbase, nn, [&]() { ++tr_rounds; });
    asm volatile("" : "+v"(w.b.lx), "+v"(w.id));
    const u64 tr2 = __builtin_amdgcn_s_memrealtime();
#else
    ploc_rounds(w, nodes, zero_parent, ni, lane, slot, hbase, nn);
#endif

    // hand-over.  (Reading the parent's word before the rounds, so that the coherent round trip hides under them, changes nothing: k_hploc_ext at 10 M
    // 0.220 vs 0.217 ms, round 3.)
    bool fast = false; u32 nL = 0, nR = 0; u64 mine = 0;
    if (owner && q != INV) {
        mine = q == R ? dep_word(1u, L, 0u) : dep_word(1u, 0u, R);
        const u64 cur = ld_agent(dep + q);
        if ((cur >> 60) == 2ull) {                    // every other contribution is in: this one completes the node, nobody else touches the word
            const u64 tot = cur + mine;
            st_agent(dep + q, 0ull);
            nL = (u32)(tot & DEP_MASK); nR = (u32)((tot >> 30) & DEP_MASK); fast = true;
        }
    }
    const bool hfast = __shfl((int)fast, hbase) != 0;
    // storeIndices (:208-218): the <= 16 survivors of a non-root range, INVALID-terminated — unless they stay in registers
    const bool left_child = __shfl((int)(q == R), hbase) != 0;        // (the owner's q and R: this range is its parent's left child)
    if (have && !w.final_ && !hfast && slot < 16) node_store_agent(recs + rec_base(left_child, tL, tR) + slot, w.id, w.rep, w.b);
    if (owner) {
        ready = false; cw.side = 0;
        if (q != INV) {
            if (fast) { cw.side = (q == R) ? 1 : 2; L = nL; R = nR; ready = true; }
            else { drain_stores(); ready = dep_arrive(dep, q, mine, L, R); }      // the wave's node / record stores are in memory before the count moves
            pc = q;
        }
    }
    cw.id = w.id; cw.rep = w.rep; cw.b = w.b;
#ifdef ABL_EXT_TRACE
    asm volatile("" : "+v"(L), "+v"(R));
    const u64 tr3 = __builtin_amdgcn_s_memrealtime();
    if (have && slot == 0) ext_trace(trace, trace_count, trace_cap, tr0, trL, trR, ni, fast ? 1u : ready ? 2u : 3u, tr_rounds, tr1, tr2, tr3);
#endif
#ifdef ABL_EXT_TIMING
    {   const u64 t3 = __builtin_amdgcn_s_memrealtime();
        const u64 hm = __ballot(have && slot == 0);
        // per-wave accumulation in the caller's registers (prof points at the wave's 16 counters, lane 0 only)
        if (lane == 0) {
            prof[0] += 1u; if (__popcll(hm) == 2) prof[1] += 1u;
            prof[2] += (u32)__popcll(hm); prof[3] += nrounds;
            prof[4] += (u32)(t2 - t1); prof[5] += (u32)(t1 - t0); prof[6] += (u32)(t3 - t2);
        }
        const bool own2 = have && slot == 0;
        const u64 mf = __ballot(own2 && fast), mc = __ballot(own2 && !fast && ready), ms = __ballot(own2 && !fast && !ready && q != INV);
        if (lane == 0) { prof[9] += (u32)__popcll(mf); prof[11] += (u32)__popcll(mc); prof[12] += (u32)__popcll(ms); }
    }
#endif
}

// External nodes (ranges crossing the tiles of k_hploc_block): the sub-queues hold the nodes whose dependencies were complete
// when the block kernel ended; every wave takes two at a time and climbs while it keeps completing parents (async_climb).
#ifndef HPX_OCC
#define HPX_OCC 6        // waves per SIMD of k_hploc_ext (80 VGPRs; round 4, after a pass's loads went out together: 10 M 0.2116 -> 0.2055 ms, 40 M 0.633 -> 0.593, 1 M / 2 M / 5 M unchanged; 7: 0.252, 8: 0.294).  Before: 4 (128 VGPRs).  Round 1: 6 (80 VGPRs; emit 0.97 ms at 10 M vs 1.00 at 7, 1.13 at 8); with ext_pass's carried
                         // survivors the kernel wants more registers: k_hploc_ext at 10 M 0.226 (6) / 0.218 (5) / 0.214 (4) / 0.216 (3) / 0.251 (7) ms, 40 M 0.719 -> 0.669, 1 M unchanged
#endif
// TICKETS (inputs >= HPX_TICKETS_MIN_N): a task owner (lanes 0 and 32) that is not climbing takes the next item of its sub-queue
// through a ticket word next to the count, so a wave that keeps climbing with one task fills its other half from the queue
// instead of running half empty.  Two-stage pipeline so that the wave never waits for it: ticket requested in one pass, item
// loaded in the next, run in the third.  Small inputs are a pure dependency chain where the static deal is faster (same box,
// k_hploc_ext us, static / tickets: 1 M 90 / 108, 2 M 102 / 116, 10 M 249 / 242, 40 M 934 / 743).
constexpr u32 HPX_TICKETS_MIN_N = 8000000u;
template <typename K, bool TICKETS>
__global__ __launch_bounds__(256, HPX_OCC) void k_hploc_ext(const bvh_aabb* __restrict__ boxes, const K* __restrict__ skeys,
                                                   const u32* __restrict__ svals, bvh_primref* leaves, bvh2_node* nodes,
                                                   bvh2_node* recs, u64* dep, u32* zero_parent,
                                                   const u32* __restrict__ q_pc, const u64* __restrict__ q_rng, u32* q_count, u32 q_cap, u32 n) {
    __shared__ u64 s_nn[256 / WAVE][WAVE];
    const int lane = tid_x() & (WAVE - 1);
    const WaveList wl{ nullptr, nullptr, nullptr, nullptr };
    const u32 nwaves = nbid_x() * (256 / WAVE);
    const u32 wid = bid_x() * (256 / WAVE) + (u32)__builtin_amdgcn_readfirstlane((int)(tid_x() >> 6));   // (wave-uniform: the sub-queue's index, base and length live in SGPRs —
    const u32 sub = wid % HPQ_SUB;                                       //  as vector values two of them were spilled to scratch at 80 VGPRs; nwaves is a multiple of HPQ_SUB)
    const u32 total = q_count[sub * 32u];
#ifdef ABL_EXT_TRACE
    u64* const trace = const_cast<u64*>(q_rng) + (size_t)q_cap * (HPQ_SUB - 1) + q_cap / 2u;      // the unused second half of the last sub-queue's storage
    u32* const trace_count = q_count + 32u + 2u;                         // word 2 of sub-queue 1's padded head
    const u32 trace_cap = (q_cap / 2u) / 4u;
    if (wid == 0u && lane == 0) { const u32 at = atomicAdd(trace_count, 1u); u64* e = trace + (size_t)at * 4u; e[0] = e[1] = __builtin_amdgcn_s_memrealtime(); e[2] = 0ull; e[3] = 0ull; }   // kernel start
#endif
#ifdef ABL_EXT_TIMING
    u32 profw[16] = { 0 };
    u32* const prof = profw;
    const u64 tw0 = __builtin_amdgcn_s_memrealtime();
    // flushed when the wave ends: per-sub-queue copies of the counters (words 3..15 of the sub-queue's padded head) + a histogram of wave lifetimes (words 16..31, 2^15 cycles per bin)
    struct WaveEnd { u32* out; u32* prof; u64 tw0; int lane; __device__ ~WaveEnd() { if (lane == 0) {
        const u32 life = (u32)(__builtin_amdgcn_s_memrealtime() - tw0);
        for (int k = 0; k < 13; ++k) if (prof[k]) atomicAdd(out + 3 + k, k >= 4 && k <= 6 ? prof[k] >> 6 : prof[k]);
        atomicAdd(out + 3 + 7, life >> 10); atomicAdd(out + 3 + 8, 1u);
        const u32 bin = life >> 15; atomicAdd(out + 16 + (bin < 15u ? bin : 15u), 1u); } } } wave_end{ q_count + sub * 32u, profw, tw0, lane };
#else
    u32* const prof = nullptr;
#endif
    if (TICKETS) {
        u32* head = q_count + sub * 32u + 1u;
        // (`pending` says that a ticket is outstanding; the ticket's VALUE — the returning atomic's result — is only looked at under it: testing the value itself at the top of
        //  every iteration made the wave wait for everything outstanding, the previous pass's write-through node stores included, before it could request the next pass's
        //  loads — round 4, found in the ISA)
        bool ready = false, dry = false, have_item = false, pending = false;
        u32 pc = 0, L = 0, R = 0, tk = 0, ipc = 0; u64 irg = 0;
        ExtCarry cw; cw.id = INV; cw.rep = INV; cw.b = box_empty(); cw.side = 0;
        while (true) {
            if (have_item && !ready) { pc = ipc; L = (u32)irg; R = (u32)(irg >> 32); ready = true; have_item = false; }
            if (pending && !have_item) {
                if (tk < total) { const size_t at = (size_t)sub * q_cap + tk; ipc = q_pc[at]; irg = q_rng[at]; have_item = true; }
                else dry = true;
                pending = false;
            }
            if ((lane & 31) == 0 && !ready && !have_item && !pending && !dry) { tk = atomicAdd(head, 1u); pending = true; }
            const u64 rm = __ballot(ready);
            if (!rm) { if (__ballot(pending || have_item)) continue; break; }
            ext_pass(ready, pc, L, R, cw, skeys, boxes, svals, leaves, nodes, recs, dep, zero_parent, n - 1, lane, s_nn[tid_x() / WAVE], wl, prof EXT_TRACE_PASS);
        }
        return;
    }
    const u32 wsub = wid / HPQ_SUB, nwsub = nwaves / HPQ_SUB;
    for (u32 base = wsub * 2u; base < total; base += nwsub * 2u) {       // wave-uniform
        const u32 idx = base + (u32)(lane >> 5);
        bool ready = (lane & 31) == 0 && idx < total;
        u32 pc = 0, L = 0, R = 0;
        if (ready) { const size_t at = (size_t)sub * q_cap + idx; pc = q_pc[at]; const u64 rg = q_rng[at]; L = (u32)rg; R = (u32)(rg >> 32); }
        ExtCarry cw; cw.id = INV; cw.rep = INV; cw.b = box_empty(); cw.side = 0;
        while (__ballot(ready)) ext_pass(ready, pc, L, R, cw, skeys, boxes, svals, leaves, nodes, recs, dep, zero_parent, n - 1, lane, s_nn[tid_x() / WAVE], wl, prof EXT_TRACE_PASS);
    }
}

// =====================================================================================================================
// Overlapped schedule (round 6; BVH_OPT_HPLOC_SCHEDULER = 3, NOT the default): the external climb runs BESIDE the tile kernel instead of behind it.
// Measured slower on the MI355X at every shape tried (10 M emit 0.886-1.20 ms against 0.750, 2 M 0.257 against 0.224; LEADS.md row 87, profiles/r06_live_timeline.md):
// the tile kernel leaves no idle issue slots for the climb to hide in — with the consumers resident it runs 0.70-0.83 ms instead of 0.56: the ~0.135 ms of
// VALU-bound climbing it absorbs (profiles/r06_ext_chain.md), and more.  Kept as a selectable schedule with identical trees (tests/test_gpu_round6.py), like the other measured-and-dropped formulations.
//
// k_hploc_live is launched on a second stream as a small resident grid (HPL_GRID workgroups of 256 threads) whose half-waves take tickets in the 64 sub-queues and
// poll their slot until the tile kernel has filled it (queue_put<LIVE>: a slot is its own flag), run the node and climb exactly like k_hploc_ext.  The tile kernel
// never waits for this kernel, so whatever the dispatcher does with the two launches there is no cycle: a consumer only ever waits for tiles, and tiles only need a
// free slot on some CU.  A consumer retires when every tile has handed over (queue_count[HPQ_DONE_WORD] == ntiles), its sub-queue's count — final by then — does not
// reach its ticket, and it is not climbing; the wave that completes the root ends the build.  The idea: only the chain above the LAST tiles stays visible.
// What crosses between the two running launches: the tiles' survivor records and queue items (write-through stores, drained before the dependency word moves),
// node 0 and the node that points at it (ploc_rounds_lds<ZERO_WT>), the dependency words (agent-scope atomics on both sides).  Leaves: see ext_pass<LIVE>.
// A watchdog on the 100 MHz clock ends a consumer that saw no progress for HPL_WATCHDOG_TICKS (a tile kernel that never came: a failed launch, a reset) — the build is
// wrong then, but the device is not hung.
// =====================================================================================================================
#ifndef HPL_OCC
#define HPL_OCC 6        // launch bound (waves per SIMD) of k_hploc_live: the register budget of k_hploc_ext
#endif
#ifndef HPL_GRID
#define HPL_GRID 256u    // workgroups (a multiple of 16: waves are dealt to the 64 sub-queues round-robin)
#endif
#ifndef HPL_SLEEP
#define HPL_SLEEP 24     // s_sleep argument of an idle wave's poll (64 clocks each)
#endif
constexpr u64 HPL_WATCHDOG_TICKS = 400000000ull;       // 4 s of the 100 MHz clock
constexpr u32 HPL_POISON = 0xFFFFFFFFu;                // a slot's pc word: "the queue ended before this ticket" (pc + 1 of a real item is < 2^30)
// Ending.  Nobody polls the "tiles done" word but ONE wave (the monitor: wave 0 of workgroup 0, which takes no tasks): 2048 half-waves polling one word saturate its
// memory channel and with it everything else that maps there — the first build of this schedule ran the tile kernel three times slower for it.  When every tile has
// handed over the sub-queues' counts are final; the monitor then poisons, in every sub-queue, as many slots behind the count as the sub-queue has consumers.  Tickets are
// handed out in order and a consumer takes a new one only after its previous one was served, so every consumer ends its life on exactly one poisoned slot, which it zeroes
// like any other: the slots are clean again when the launch ends.
__device__ __forceinline__ u32 live_consumers(u32 nwaves, u32 sub) { return 2u * (nwaves / HPQ_SUB - (sub == 0u ? 1u : 0u)); }
template <typename K>
__global__ __launch_bounds__(256, HPL_OCC) void k_hploc_live(const bvh_aabb* __restrict__ boxes, const K* __restrict__ skeys, const u32* __restrict__ svals,
                                                    bvh2_node* nodes, bvh2_node* recs, u64* dep, u32* zero_parent,
                                                    u32* q_pc, u64* q_rng, u32* q_count, u32 q_cap, u32 n, u32 ntiles) {
    __shared__ u64 s_nn[256 / WAVE][WAVE];
    const int lane = tid_x() & (WAVE - 1);
    const WaveList wl{ nullptr, nullptr, nullptr, nullptr };
    const u32 nwaves = nbid_x() * (256 / WAVE);
    const u32 wid = bid_x() * (256 / WAVE) + (u32)__builtin_amdgcn_readfirstlane((int)(tid_x() >> 6));
    u64 t_last = __builtin_amdgcn_s_memrealtime();
    if (wid == 0u) {     // the monitor
        while (ld_agent(q_count + HPQ_DONE_WORD) != ntiles) {
            if (__builtin_amdgcn_s_memrealtime() - t_last > HPL_WATCHDOG_TICKS) break;
            __builtin_amdgcn_s_sleep(8);
        }
        drain_stores();                                                       // (the counts are requested after "all tiles done" has arrived: they are final)
        const u32 sub = (u32)lane;                                            // HPQ_SUB == WAVE: a sub-queue per lane
        const u32 c = ld_agent(q_count + sub * 32u), m = live_consumers(nwaves, sub);
#ifdef BVH_ABLATION      // measurement build: items queued and tickets taken at the moment the last tile handed over (words 1 / 2 behind the "tiles done" word; tools/ab_live.py)
        atomicAdd(q_count + HPQ_DONE_WORD + 1u, c); atomicAdd(q_count + HPQ_DONE_WORD + 2u, ld_agent(q_count + sub * 32u + 1u));
#endif
        for (u32 k = 0; k < m; ++k) if (c + k < q_cap) st_agent(q_pc + (size_t)sub * q_cap + c + k, HPL_POISON);
        return;
    }
    static_assert(HPQ_SUB == (u32)WAVE, "the monitor deals one sub-queue to each lane");
    const u32 sub = wid % HPQ_SUB;
    u32* const head = q_count + sub * 32u + 1u;
    const bool owner_lane = (lane & 31) == 0;
    bool ready = false, dry = false, ticket = false;
    u32 pc = 0, L = 0, R = 0, tk = 0, ipc = 0;
    ExtCarry cw; cw.id = INV; cw.rep = INV; cw.b = box_empty(); cw.side = 0;
    while (true) {
        if (owner_lane && !ready && !dry) {
            if (!ticket) { tk = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ticket = true; ipc = 0u; }
            if (tk >= q_cap) dry = true;                                      // (cannot happen: the capacity covers every item a build can queue, and then some)
            else {
                const size_t at = (size_t)sub * q_cap + tk;
                if (ipc == 0u) ipc = ld_agent(q_pc + at);
                if (ipc == HPL_POISON) { st_agent(q_pc + at, 0u); dry = true; }
                else if (ipc != 0u) {
                    const u64 irg = ld_agent(q_rng + at);                     // (the item's two words are independent stores: both must have landed)
                    if (irg != 0ull) {
                        st_agent(q_pc + at, 0u); st_agent(q_rng + at, 0ull);  // the slot is clean for the next build
                        pc = ipc - 1u; L = (u32)irg; R = (u32)(irg >> 32); ready = true; ticket = false;
                    }
                }
            }
        }
        const u64 rm = __ballot(ready);
        if (!rm) {
            if (!__ballot(owner_lane && !dry)) break;
            if (__builtin_amdgcn_s_memrealtime() - t_last > HPL_WATCHDOG_TICKS) break;
            __builtin_amdgcn_s_sleep(HPL_SLEEP);
            continue;
        }
        ext_pass<K, true>(ready, pc, L, R, cw, skeys, boxes, svals, nullptr, nodes, recs, dep, zero_parent, n - 1, lane, s_nn[tid_x() / WAVE], wl);
        t_last = __builtin_amdgcn_s_memrealtime();
    }
}

void launch_hploc(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                  void* d_nodes, void* d_leaves, const HplocScratch& sc) {
    const u32 gaps = n - 1;
    const dim3 g((gaps + HP_BLOCK - 1) / HP_BLOCK), b(HP_BLOCK);
    KernelScope ks(s, "k_hploc");
    if (key_bits == 64) hipLaunchKernelGGL(k_hploc<u64>, g, b, 0, s, (const bvh_aabb*)d_boxes, (const u64*)d_skeys, d_svals, (bvh_primref*)d_leaves, (bvh2_node*)d_nodes, (bvh2_node*)sc.recs, sc.dep, sc.zero_parent, n, hploc_ablation());
    else                hipLaunchKernelGGL(k_hploc<u32>, g, b, 0, s, (const bvh_aabb*)d_boxes, (const u32*)d_skeys, d_svals, (bvh_primref*)d_leaves, (bvh2_node*)d_nodes, (bvh2_node*)sc.recs, sc.dep, sc.zero_parent, n, hploc_ablation());
}

// Block-local HPLOC for large n (n > 2 tiles: the root is never local).  Tile 512 leaves / 256 threads / 7 waves per SIMD (72 VGPRs,
// 3 spilled) measured best on MI355X with the LDS-atomic neighbour selection: 10 M emit 0.94 ms vs 0.97 at 6 waves.
#ifndef HPB_T
#define HPB_T 512
#endif
#ifndef HPB_NT
#define HPB_NT 256
#endif
#ifndef HPB_OCC
#define HPB_OCC 8        // waves per SIMD = workgroups per CU of the tile kernel (HPB_LEAN: 8 x 20.3 KB of LDS, 64 VGPRs)
#endif
#ifndef HPB_OCC_1024
#define HPB_OCC_1024 6    // (the 1024-leaf / 512-thread shape of the A/B builds)
#endif
#ifndef HPB_OCC64
#define HPB_OCC64 7      // the u64-key instantiation keeps more registers (96-bit augmented keys, wider key window)
#endif
static void hpb_config(int* t, int* nt, int* occ) {
    *t = HPB_T; *nt = HPB_NT; *occ = HPB_OCC;
#ifdef BVH_ABLATION
    const char* e = getenv("BVH_HPB");                 // "T,NT,OCC" (measurements only; 1024,512 is the other compiled tile)
    if (e) { int a = 0, b = 0, c = 0; if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) { *t = a; *nt = b; *occ = c; } }
#endif
}
uint32_t hploc_block_tile() { int t, nt, occ; hpb_config(&t, &nt, &occ); return (uint32_t)t; }
// every tile may queue up to 2T nodes (its own external nodes + the parents of its maximal local ones), tiles >= 128 leaves
uint32_t hploc_head_words() { return HPQ_HEAD_WORDS; }
size_t hploc_queue_capacity(uint32_t n) { return (((size_t)n / 128 + 1) / HPQ_SUB + 6) * 2 * 128 * HPQ_SUB; }

void launch_hploc_block(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                        void* d_nodes, void* d_leaves, const HplocScratch& sc, bool heads_cleared, const HplocLive* live) {
    if (!heads_cleared) (void)hipMemsetAsync(sc.queue_count, 0, HPQ_HEAD_WORDS * sizeof(u32), s);
    int t, nt, occ; hpb_config(&t, &nt, &occ);
    const int dbg = hploc_ablation();
    const u32 q_cap = (u32)(sc.queue_capacity / HPQ_SUB);
#define HPB_LAUNCH(KK, TT, NN, OO, LL) hipLaunchKernelGGL((k_hploc_block<KK, TT, NN, OO, LL>), dim3((n + TT - 1) / TT), dim3(NN), 0, s, (const bvh_aabb*)d_boxes, (const KK*)d_skeys, d_svals, \
                       (bvh_primref*)d_leaves, (bvh2_node*)d_nodes, (bvh2_node*)sc.recs, sc.dep, sc.zero_parent, sc.queue_pc, sc.queue_rng, sc.queue_count, q_cap, n, dbg, (const float4*)sc.leaf_tris)
    if (live && live->side && !dbg && !(t == 1024 && nt == 512)) {
        // overlapped schedule: the consumer grid goes to the side stream behind everything enqueued on s so far (the sort), the tile kernel to s; s then waits for the consumer
        const u32 ntiles = (n + (u32)HPB_T - 1u) / (u32)HPB_T;
        KernelScope ks(s, "k_hploc_block");
        if (hipEventRecord(live->fork, s) != hipSuccess || hipStreamWaitEvent(live->side, live->fork, 0) != hipSuccess) return;
        // the tile kernel first: a consumer grid whose producer was refused would only end by its watchdog
        if (key_bits == 64) HPB_LAUNCH(u64, HPB_T, HPB_NT, HPB_OCC64, true); else HPB_LAUNCH(u32, HPB_T, HPB_NT, HPB_OCC, true);
        if (hipPeekAtLastError() != hipSuccess) return;
#define HPL_LAUNCH(KK) hipLaunchKernelGGL((k_hploc_live<KK>), dim3(HPL_GRID), dim3(256), 0, live->side, (const bvh_aabb*)d_boxes, (const KK*)d_skeys, d_svals, \
                       (bvh2_node*)d_nodes, (bvh2_node*)sc.recs, sc.dep, sc.zero_parent, sc.queue_pc, sc.queue_rng, sc.queue_count, q_cap, n, ntiles)
        if (key_bits == 64) HPL_LAUNCH(u64); else HPL_LAUNCH(u32);
#undef HPL_LAUNCH
        (void)hipEventRecord(live->join, live->side);
        KernelScope ks2(s, "k_hploc_live(tail)");      // (per-kernel events: from the tile kernel's end to the consumer's)
        (void)hipStreamWaitEvent(s, live->join, 0);
        return;
    }
    { KernelScope ks(s, "k_hploc_block");
      if (key_bits == 64) HPB_LAUNCH(u64, HPB_T, HPB_NT, HPB_OCC64, false);
      else if (t == 1024 && nt == 512) HPB_LAUNCH(u32, 1024, 512, HPB_OCC_1024, false);
      else HPB_LAUNCH(u32, HPB_T, HPB_NT, HPB_OCC, false); }
#undef HPB_LAUNCH
    if (dbg) return;
    KernelScope ks(s, "k_hploc_ext");
#ifndef HPX_GRID
#define HPX_GRID 2048u
#endif
    const u32 xg = HPX_GRID;                            // (measured flat from 2048 to 8192 workgroups; at HPX_OCC = 6, 1536 are resident); a multiple of 16:
                                                        // waves are dealt to the 64 sub-queues round-robin
#define HPX_LAUNCH(KK, TT) hipLaunchKernelGGL((k_hploc_ext<KK, TT>), dim3(xg), dim3(256), 0, s, (const bvh_aabb*)d_boxes, (const KK*)d_skeys, d_svals, (bvh_primref*)d_leaves, \
                       (bvh2_node*)d_nodes, (bvh2_node*)sc.recs, sc.dep, sc.zero_parent, (const u32*)sc.queue_pc, (const u64*)sc.queue_rng, sc.queue_count, q_cap, n)
    const bool tickets = n >= HPX_TICKETS_MIN_N;
    if (key_bits == 64) { if (tickets) HPX_LAUNCH(u64, true); else HPX_LAUNCH(u64, false); }
    else                { if (tickets) HPX_LAUNCH(u32, true); else HPX_LAUNCH(u32, false); }
#undef HPX_LAUNCH
}

// (kernels.hpp: touching one kernel of this translation unit makes the runtime load its code object — bvh_ctx_create does that for the build path's modules, so
// that a context's FIRST build does not pay for it: 0.3-0.7 ms per module on the MI355X, tools/cold_probe.py)
void warm_hploc() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_hploc<u32>)); }

} // namespace bvh
