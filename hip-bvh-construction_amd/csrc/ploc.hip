// ploc.hip — stage B for PLOC++ on gfx950.
//
// Replaces SetupClusters + Ploc + SinglePassPloc and the host iteration loop of the reference (src/Ploc++Kernel.h:39-55,
// :211-362, :98-209; src/PLOC++Bvh.cpp:82-152).  Output: Bvh2Node[n-1] (root = node 0) + PrimRef[n] leaves.
//
// One iteration = for every cluster of the Morton-ordered list find the nearest neighbour within +-8 list positions under
// the {area-of-union bits, list position} order, merge mutual pairs (lower position owns the new node), compact the list
// in order.  The reference evaluates this per 1024-cluster chunk with a 16-entry halo on both sides (:232-270), which is
// what this kernel does too; chunks are chained with decoupled look-back on one 64-bit status word {flag, merges, kept}
// instead of the reference's serial spin chain over blocks (:341-347), and the new node index is C-2-(rank of the merge in
// list order) — one of the arrival orders the reference's global atomicAdd (:311) can produce, here deterministic.
// The iteration loop stays on the device: iteration k reads the cluster count from state[k], writes state[k+1]; the host
// enqueues a batch of launches without reading anything back (the reference syncs D2H every iteration, src/PLOC++Bvh.cpp:150).
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

#ifndef PLOC_NARROW
#define PLOC_NARROW 512
#endif
#ifndef PLOC_NN_OWN_F64
#define PLOC_NN_OWN_F64 1   // 1 (round 4): nn_pairs keeps an entry's own candidates in registers (v_min_f64 on the 64-bit key) and sends one atomic per entry; 0: both ends of every pair by LDS atomics
#endif
#ifndef PLOC_TAIL_PAIRS
#define PLOC_TAIL_PAIRS 1
#endif
#ifndef PLOC_ONE_SHOT_MAX_N
#define PLOC_ONE_SHOT_MAX_N (1 << 20)
#endif
#ifndef PLOC_ABL
#define PLOC_ABL 0       // measurements only (tools/build_variant.sh): 1 no look-back wait, 2 no NN search, 3 no list stores — results are wrong
#endif
// workgroup size: 512 threads (2 clusters per thread, 4 workgroups = 32 waves per CU) for the early, throughput-bound iterations;
// 1024 threads (one cluster per thread, the reference's shape) for the late ones, where an iteration is a handful of chunks and
// its time is one chunk's critical path (measured at 10 M: 22 us per late iteration with 256 threads, 8 us with 1024).  Measured
// at 10 M (whole emit): 128 threads 2.77 ms, 256: 2.20, 512: 2.03, 1024 throughout: 2.22.  Ablations of the first iteration (352 us
// with 256 threads): waiting for the predecessors' totals in the look-back 131 us, list stores 71 us, nearest-neighbour search 55 us;
// evaluating each pair once (rows of 56 owned positions + __shfl_up) and a workgroup-wide look-back walk were both slower.
constexpr int PL_RADIUS = 8;                   // PlocRadius, src/Common.h:595
constexpr int PL_HALO = 2 * PL_RADIUS;         // :219-221
constexpr int PL_SPAN = PLOC_CHUNK + 2 * PL_HALO;
constexpr u64 PS_LOCAL = 1ull << 62, PS_INCL = 2ull << 62;

struct PlocLds {
    float lx[PL_SPAN], ly[PL_SPAN], lz[PL_SPAN], hx[PL_SPAN], hy[PL_SPAN], hz[PL_SPAN];
    u32 id[PL_SPAN];
    u64 nn[PL_SPAN];        // nearest neighbour key {area bits, index into the span arrays}; the index is the low word
    u32 wsum[1024 / WAVE];
    u32 bcast[4];
};

__device__ __forceinline__ Box lds_box(const PlocLds& s, int k) { return { s.lx[k], s.ly[k], s.lz[k], s.hx[k], s.hy[k], s.hz[k] }; }
__device__ __forceinline__ void lds_set(PlocLds& s, int k, u32 id, const Box& b) {
    s.id[k] = id; s.lx[k] = b.lx; s.ly[k] = b.ly; s.lz[k] = b.lz; s.hx[k] = b.hx; s.hy[k] = b.hy; s.hz[k] = b.hz;
}
// The cluster list carries the boxes: one 32-byte entry {id, box, pad} per cluster, read and written as two float4.  The
// reference keeps ids only and gathers every cluster's box from the leaf / node arrays in every iteration (:237-241) — on
// HBM a random 24..32-byte gather costs a full sector and a dependent round trip; streaming 32 B/cluster does not.
__device__ __forceinline__ void entry_load(const float4* __restrict__ list, size_t g, u32& id, Box& b) {
    const float4 a = list[2 * g], c = list[2 * g + 1];
    id = __float_as_uint(a.x); b = { a.y, a.z, a.w, c.x, c.y, c.z };
}
__device__ __forceinline__ void entry_store(float4* __restrict__ list, size_t g, u32 id, const Box& b) {
    list[2 * g] = make_float4(__uint_as_float(id), b.lx, b.ly, b.lz);
    list[2 * g + 1] = make_float4(b.hx, b.hy, b.hz, 0.0f);
}

// nearest neighbour of span entry k among valid entries [lo, hi) within +-8, key {area bits, position}
__device__ __forceinline__ u32 nearest(const PlocLds& s, int k, int lo, int hi) {
    const Box b = lds_box(s, k);
    u64 best = ~0ull;
#pragma unroll
    for (int r = 1; r <= PL_RADIUS; ++r) {
        const int a = k - r, c = k + r;
        if (a >= lo) { const u64 key = ((u64)__float_as_uint(box_area(box_union(lds_box(s, a), b))) << 32) | (u32)a; best = key < best ? key : best; }
        if (c < hi)  { const u64 key = ((u64)__float_as_uint(box_area(box_union(lds_box(s, c), b))) << 32) | (u32)c; best = key < best ? key : best; }
    }
    return (u32)best;
}

// block-wide exclusive scan of a packed {merges<<16 | kept} per-thread count; returns exclusive prefix, total via *total
template <int PL_BLOCK>
__device__ __forceinline__ u32 block_scan(PlocLds& s, u32 v, u32* total) {
    const int lane = tid_x() & (WAVE - 1), wave = tid_x() / WAVE;
    u32 inc = v;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { const u32 t = (u32)__shfl_up((int)inc, off); if (lane >= off) inc += t; }
    __syncthreads();                               // protects wsum reuse
    if (lane == WAVE - 1) s.wsum[wave] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < PL_BLOCK / WAVE; ++w) { const u32 c = s.wsum[w]; if (w < wave) base += c; tot += c; }
    *total = tot;
    return base + inc - v;
}

// The last rounds of the single-workgroup tail: once <= 64 clusters are left they sit one per lane in ONE wave and the remaining ~19 rounds (the list
// shrinks ~20 % per round) run without a barrier — a round of the workgroup loop above costs ~2.7 us (four barriers, a block scan, 16 neighbour boxes
// read from LDS per cluster), a wave round ~0.8 us.  Same rule: nearest neighbour within +-8 positions under {area bits, position}, mutual pairs merge,
// the lower position owns the node, node index = c - 2 - (merges at lower positions), survivors keep their order (SinglePassPloc :131-205).
// Neighbour boxes come through a DPP wave_shl:1 chain; every pair is evaluated once and minimised into both ends' key words (LDS atomics; a wave's
// LDS operations execute in order); compaction goes through the LDS list (write to the rank, read the own position back).
// Round 3, measured at Sponza-262 144: tail launch 75 -> 58 us, emit 0.383 -> 0.366 ms.  (Also tried: late launches with LDS for 4096 clusters so that the
// single-workgroup tail starts below 4096 instead of 1024 — the ten 8-us launches it replaces become six in-LDS rounds of ~5 us on ONE CU plus more empty
// launches at the end of the batch: 0.386 ms, dropped.)
template <bool OWN>      // OWN: the lane's own candidates (lane, lane + r) in a register pair, v_min_f64 on the 64-bit key (nn_pairs_fn)
__device__ __forceinline__ void ploc_tail_wave(PlocLds& s, u32 c, bvh2_node* __restrict__ nodes, int lane) {
    u32 id = (u32)lane < c ? s.id[lane] : INV;
    Box b = (u32)lane < c ? lds_box(s, lane) : box_empty();
    const u64 lt = lanemask_lt();
    while (c > 1u) {
        s.nn[lane] = ~0ull; s.nn[lane + WAVE] = ~0ull;     // (pairs reach at most 8 positions beyond the last lane)
        compiler_fence();
        Box nb = b;
        double own = __longlong_as_double(0x7FEFFFFFFFFFFFFFll);
#pragma unroll
        for (int r = 1; r <= PL_RADIUS; ++r) {
            nb = box_shl1(nb);                             // box of position lane + r
            if ((u32)(lane + r) < c) {
                const unsigned long long key = (unsigned long long)__float_as_uint(box_area(box_union(nb, b))) << 32;
                atomicMin(reinterpret_cast<unsigned long long*>(s.nn + lane + r), key | (u32)lane);
                if constexpr (OWN) own = __builtin_fmin(own, __longlong_as_double((long long)(key | (u32)(lane + r))));
                else atomicMin(reinterpret_cast<unsigned long long*>(s.nn + lane), key | (u32)(lane + r));
            }
        }
        compiler_fence();
        const bool in = (u32)lane < c;
        const u64 left = s.nn[lane], right = OWN ? (u64)__double_as_longlong(own) : ~0ull;
        const int nbr = in ? (int)(u32)(left < right ? left : right) : lane;
        const bool mutual = in && (u32)__shfl(nbr, nbr) == (u32)lane;
        const bool merge = mutual && lane < nbr, absorbed = mutual && lane > nbr;
        const u32 id_nb = (u32)__shfl((int)id, nbr);
        const Box bn = shfl_box(b, nbr);
        const u64 mm = __ballot(merge);
        if (merge) {
            b = box_union(b, bn);
            const u32 at = c - 2u - (u32)__popcll(mm & lt);                     // :168
            node_store_plain(nodes + at, id, id_nb, b);
            id = at;
        }
        const bool keep = in && !absorbed;
        const u64 km = __ballot(keep);
        if (keep) lds_set(s, (int)__popcll(km & lt), id, b);
        c = (u32)__popcll(km);
        id = (u32)lane < c ? s.id[lane] : INV;
        b = (u32)lane < c ? lds_box(s, lane) : box_empty();
    }
}

// Nearest neighbours of the span entries [lo, hi) (lo: first valid entry, hi: one past the last): everything a chunk cluster or its neighbour needs (:252-270).  As the
// reference does it: every pair (s, s + r), r = 1..8, is evaluated ONCE and minimised as {area bits, other position} into both ends' words (LDS atomics).
template <int PL_BLOCK>
__device__ __forceinline__ void nn_pairs_fn(PlocLds& s, const int tid, const int lo, const int hi) {
        // A 16-lane DPP row holds 32 consecutive span entries, two per lane, and evaluates the pairs of its first 24 (common.hpp, row_shl): the
        // boxes are read from LDS once (round 1 read the 16 neighbours of every entry: 192 LDS reads per cluster, the limiter of the mid-size iterations).
        {
            constexpr int ROWS = PL_BLOCK / 16, TILES = (PL_HALO + PLOC_CHUNK + PL_RADIUS + 23) / 24;     // pairs with s in [0, 1048)
            static_assert(TILES % 4 == 0 || TILES <= ROWS, "whole waves per pass");
            const int row = tid >> 4, rl = tid & 15;
            for (int t = row; t < TILES && 24 * t < hi; t += ROWS) {
                const int sA = 24 * t + 2 * rl, sB = sA + 1;
                const Box bA = (sA >= lo && sA < hi) ? lds_box(s, sA) : box_empty(), bB = (sB >= lo && sB < hi) ? lds_box(s, sB) : box_empty();
                const int limA = (rl < 12 && sA >= lo) ? hi - sA : 0, limB = (rl < 12 && sB >= lo) ? hi - sB : 0;   // pair (s, s + r) exists iff r < lim
                // an entry's own candidates (s, s + r): running minimum of the same 64-bit key in a register pair — one v_min_f64 each (a non-negative f32 area as the
                // high word makes the key a non-negative finite f64, which orders like its bit pattern; hploc.hip, HP_NN_LDS = 3) — and ONE atomic at the end
                // (the entry's word also collects the keys of the pairs (s - r, s), from this and from other rows): 18 LDS atomics per lane and tile instead of 32.
                // Both instantiations since round 6: the 512-thread one spilled the two accumulators at 96 VGPRs (five waves per SIMD: 10 M 2.158 -> 2.204 ms); compiled for four
                // (PLOC_OCC = 4: 106 VGPRs, no scratch) it gains (same box, LEADS.md row 96).  262 144 (1024-thread shape only) 0.3744 -> 0.3693 ms in round 5.
                constexpr bool OWN = PLOC_NN_OWN_F64 != 0;
                double ownA = __longlong_as_double(0x7FEFFFFFFFFFFFFFll), ownB = ownA;
                auto cand = [&](const Box& nA, const Box& nB, const int r) {
                    const v2f_t lx = { fminf(nA.lx, bA.lx), fminf(nB.lx, bB.lx) }, ly = { fminf(nA.ly, bA.ly), fminf(nB.ly, bB.ly) }, lz = { fminf(nA.lz, bA.lz), fminf(nB.lz, bB.lz) };
                    const v2f_t hx = { fmaxf(nA.hx, bA.hx), fmaxf(nB.hx, bB.hx) }, hy = { fmaxf(nA.hy, bA.hy), fmaxf(nB.hy, bB.hy) }, hz = { fmaxf(nA.hz, bA.hz), fmaxf(nB.hz, bB.hz) };
                    const v2f_t area = area_pair(lx, ly, lz, hx, hy, hz);
                    const unsigned long long kA = (unsigned long long)__float_as_uint(area.x) << 32, kB = (unsigned long long)__float_as_uint(area.y) << 32;
                    if (r < limA) {
                        atomicMin(reinterpret_cast<unsigned long long*>(s.nn + sA + r), kA | (u32)sA);
                        if constexpr (OWN) ownA = __builtin_fmin(ownA, __longlong_as_double((long long)(kA | (u32)(sA + r))));
                        else atomicMin(reinterpret_cast<unsigned long long*>(s.nn + sA), kA | (u32)(sA + r));
                    }
                    if (r < limB) {
                        atomicMin(reinterpret_cast<unsigned long long*>(s.nn + sB + r), kB | (u32)sB);
                        if constexpr (OWN) ownB = __builtin_fmin(ownB, __longlong_as_double((long long)(kB | (u32)(sB + r))));
                        else atomicMin(reinterpret_cast<unsigned long long*>(s.nn + sB), kB | (u32)(sB + r));
                    }
                };
                cand(bB, row_shl<1>(bA), 1);               cand(row_shl<1>(bA), row_shl<1>(bB), 2);
                cand(row_shl<1>(bB), row_shl<2>(bA), 3);   cand(row_shl<2>(bA), row_shl<2>(bB), 4);
                cand(row_shl<2>(bB), row_shl<3>(bA), 5);   cand(row_shl<3>(bA), row_shl<3>(bB), 6);
                cand(row_shl<3>(bB), row_shl<4>(bA), 7);   cand(row_shl<4>(bA), row_shl<4>(bB), 8);
                if constexpr (OWN) {
                    if (1 < limA) atomicMin(reinterpret_cast<unsigned long long*>(s.nn + sA), (unsigned long long)__double_as_longlong(ownA));
                    if (1 < limB) atomicMin(reinterpret_cast<unsigned long long*>(s.nn + sB), (unsigned long long)__double_as_longlong(ownB));
                }
            }
        }
}

// counts[k] = cluster count at the start of iteration k; tickets[k] = chunk ticket of iteration k; status: u64 per chunk
// FIRST: the build's first iteration reads the clusters straight from the sorted values and the primitive boxes and writes the
// PrimRef leaves on the way — SetupClusters (:39-55) fused: the initial cluster list (32 B written + 32 B read per primitive) never exists.
#ifndef PLOC_OCC
#define PLOC_OCC 4       // waves per SIMD the 512-thread instantiation is compiled for (5 until round 6: 96 VGPRs)
#endif
#ifndef PLOC_STATIC_G
#define PLOC_STATIC_G 256     // largest grid that takes static chunk ids (0: tickets always)
#endif
// One iteration as workgroup `wg` of `G`.  Returns true when the build is over (one cluster left, or this workgroup has nothing more to do in it).
template <int PL_BLOCK, bool FIRST>
__device__ __forceinline__ bool ploc_iter_body(PlocLds& s, const float4* __restrict__ list_in, float4* __restrict__ list_out, bvh2_node* __restrict__ nodes,
                                               u64* status, u32* counts, u32* tickets, u32* iters_done, u32 ni,
                                               const bvh_aabb* __restrict__ boxes, const u32* __restrict__ svals, bvh_primref* __restrict__ leaves,
                                               const u32 wg, const u32 G, const bool static_ok) {
    constexpr int PL_CPT = PLOC_CHUNK / PL_BLOCK;  // clusters per thread in the merge phase
    constexpr int SPT = (PL_SPAN + PL_BLOCK - 1) / PL_BLOCK;
    const int tid = tid_x();
    // Static chunk ids (round 5).  A grid of at most PLOC_STATIC_G workgroups is resident as a whole (one 1024-thread workgroup per CU at most), so chunk = workgroup id
    // needs no ticket to keep the look-back free of deadlock — and the chunk's list entries can be requested BEFORE the cluster count is known: the count (a scalar
    // load of what the previous launch wrote), the ticket (a returning atomic + two barriers) and the entries were three dependent round trips at the start of every
    // iteration's ~7 us.  Used when the grid turns out to cover the iteration's chunks (G >= chunks: every late iteration, every iteration below 256 chunks); the
    // speculative entries of positions beyond the count are masked like any halo position.  Otherwise the ticket path runs unchanged.
    u32 sid_[SPT]; Box sb_[SPT];
    // static_ok: the host's word that this launch has the device to itself (api.hip: BVH_OPT_PLOC_SCHEDULER 0).  The look-back below spins on the predecessors' status words
    // and has no helping path: with tickets a predecessor chunk is always held by a workgroup that is already running; with static ids it is held by a LOWER workgroup id,
    // which the hardware dispatches first — true on this chip, but nothing a host that shares the device (the batched builder's lanes, CU masks) should have to rely on.
    const bool spec = !FIRST && PL_BLOCK == 1024 && static_ok && G <= (u32)PLOC_STATIC_G;       // (the 512-thread shape serves the early iterations of large inputs: thousands of chunks)
    if (spec) {
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const int k = tid + q * PL_BLOCK;
            const long long gpos = (long long)wg * PLOC_CHUNK - PL_HALO + k;
            const size_t gc = (k < PL_SPAN && gpos >= 0 && gpos <= (long long)ni) ? (size_t)gpos : (size_t)wg * PLOC_CHUNK;      // (inside the list's n entries: G <= ceil(n / 1024))
            entry_load(list_in, gc, sid_[q], sb_[q]);
        }
    }
    const u32 C = counts[0];
    auto set_count = [&](u32 v) { counts[1] = v; };
    if (C <= 1) { if (wg == 0 && tid == 0) set_count(C); return true; }
    // cluster at list position g: from the list, or (FIRST) leaf g itself; own = g belongs to this chunk (not its halo): write the PrimRef
    auto fetch = [&](size_t g, bool own, u32& id, Box& b) {
        if (FIRST) {
            const u32 prim = svals[g];
            b = box_gather(boxes + prim); id = (u32)g + ni;
            if (own) {
                float* f = reinterpret_cast<float*>(leaves + g);
                reinterpret_cast<u32*>(f)[0] = prim;
                f[1] = b.lx; f[2] = b.ly; f[3] = b.lz; f[4] = b.hx; f[5] = b.hy; f[6] = b.hz;
            }
        } else entry_load(list_in, g, id, b);
    };

    auto nn_pairs = [&](const int lo, const int hi) { nn_pairs_fn<PL_BLOCK>(s, tid, lo, hi); };
    if (C < (u32)PLOC_CHUNK) {
        // ---- tail: the whole list in one workgroup until a single cluster remains (SinglePassPloc :98-209)
        if (wg != 0) return true;
        for (int k = tid; k < (int)C; k += PL_BLOCK) { u32 id; Box b; fetch((size_t)k, true, id, b); lds_set(s, k, id, b); }
        __syncthreads();
        u32 c = C;
        while (c > 1) {
            if (c <= (u32)WAVE) { if (tid < WAVE) ploc_tail_wave<PLOC_NN_OWN_F64 && PL_BLOCK == 1024>(s, c, nodes, tid); break; }         // (block-uniform; the list in LDS is complete: barrier above / at the loop's end)
#if PLOC_TAIL_PAIRS
            for (int k = tid; k < (int)c; k += PL_BLOCK) s.nn[k] = ~0ull;                             // :131-148, range clipped to [0,c): every pair once, as in the
            __syncthreads();                                                                        // iterations (nearest() reads 16 neighbour boxes per cluster from LDS)
            nn_pairs(0, (int)c);
            __syncthreads();
#else
            for (int k = tid; k < (int)c; k += PL_BLOCK) s.nn[k] = (u64)nearest(s, k, 0, (int)c);   // :131-148, range clipped to [0,c)
            __syncthreads();
#endif
            // each thread owns PL_CPT consecutive list positions
            u32 cid[PL_CPT]; Box cb[PL_CPT]; bool mrg[PL_CPT], keep[PL_CPT]; u32 pid[PL_CPT];
            u32 packed = 0;
#pragma unroll
            for (int q = 0; q < PL_CPT; ++q) {
                const int k = tid * PL_CPT + q;
                mrg[q] = false; keep[q] = false; cid[q] = INV; pid[q] = INV; cb[q] = box_empty();
                if (k < (int)c) {
                    const u32 nb = (u32)s.nn[k];
                    const bool mutual = (u32)s.nn[nb] == (u32)k;
                    mrg[q] = mutual && (u32)k < nb; keep[q] = !mutual || mrg[q];
                    cid[q] = s.id[k]; cb[q] = lds_box(s, k);
                    if (mrg[q]) { pid[q] = s.id[nb]; cb[q] = box_union(cb[q], lds_box(s, (int)nb)); }
                    packed += ((u32)mrg[q] << 16) + (u32)keep[q];
                }
            }
            u32 tot; u32 ex = block_scan<PL_BLOCK>(s, packed, &tot);     // (block_scan's barriers also order the reads above)
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PL_CPT; ++q) {
                if (keep[q]) {
                    u32 id = cid[q];
                    if (mrg[q]) {
                        id = c - 2 - (ex >> 16);                  // :168
                        node_store_plain(nodes + id, cid[q], pid[q], cb[q]);
                    }
                    lds_set(s, (int)(ex & 0xFFFFu), id, cb[q]);
                }
                ex += ((u32)mrg[q] << 16) + (u32)keep[q];
            }
            __syncthreads();
            c = tot & 0xFFFFu;
        }
        if (tid == 0) { set_count(1u); atomicAdd(iters_done, 1u); }
        return true;
    }

    // ---- one global iteration (Ploc :211-362), persistent over chunk tickets.  A chunk's compacted output needs the totals of all
    // chunks before it; instead of waiting for them the workgroup publishes its own totals, keeps the chunk's results in registers
    // and takes the next chunk — the walk over the predecessors' totals and the stores happen one chunk later, when
    // those totals have long been published.
    const u32 chunks = (C + PLOC_CHUNK - 1) / PLOC_CHUNK;
    bool p_have = false; u32 p_chunk = 0, p_tot = 0, p_ex = 0;
    u32 p_cid[PL_CPT], p_pid[PL_CPT]; Box p_cb[PL_CPT]; bool p_mrg[PL_CPT], p_keep[PL_CPT];
    // prefix of chunk `chunk` (wave 0 walks back), then the chunk's stores
    auto finish = [&](u32 chunk, u32 tot, u32 ex, const u32* cid, const u32* pid, const Box* cb, const bool* mrg, const bool* keep) {
        // chain the chunk totals: status word {flag:2, merges:31, kept:31}.  Wave 0 walks back 64 predecessors per step
        // (one load per lane, ballot for the nearest inclusive prefix, wave reduction of the aggregates in front of it) —
        // a one-thread walk costs a memory round trip per predecessor, which dominated small scenes.
        if (tid < WAVE) {
            const u64 mine = ((u64)(tot >> 16) << 31) | (u64)(tot & 0xFFFFu);
            u64 excl = 0;
            if (chunk != 0 && PLOC_ABL != 1) {
                long long hi = (long long)chunk - 1;                       // nearest predecessor not yet accounted for
                while (true) {
                    const long long idx = hi - tid;                        // lane 0 looks at the nearest
                    const u64 st = idx >= 0 ? __hip_atomic_load(status + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : PS_INCL;
                    const u64 flag = st >> 62;
                    const u64 unpub = __ballot(flag == 0), incl = __ballot(flag == 2);
                    // usable prefix of the window: lanes below the first unpublished one, up to and including the first inclusive one
                    const int first_unpub = unpub ? __ffsll((unsigned long long)unpub) - 1 : WAVE;
                    const int first_incl = incl ? __ffsll((unsigned long long)incl) - 1 : WAVE;
                    const int take = first_incl < first_unpub ? first_incl + 1 : first_unpub;      // lanes [0, take)
                    u64 v = (tid < take) ? (st & ((1ull << 62) - 1ull)) : 0ull;
#pragma unroll
                    for (int m = 1; m < WAVE; m <<= 1) v += __shfl_xor(v, m);
                    excl += v;
                    if (first_incl < first_unpub) break;
                    hi -= take;
                    if (take == 0) __builtin_amdgcn_s_sleep(1);
                }
                if (tid == 0) __hip_atomic_store(status + chunk, PS_INCL | (excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) {
                s.bcast[1] = (u32)(excl >> 31); s.bcast[2] = (u32)(excl & 0x7FFFFFFFull);
                if (chunk == chunks - 1) {                                                                                   // src/PLOC++Bvh.cpp:150
                    const u32 left = C - ((u32)(excl >> 31) + (tot >> 16));
                    counts[1] = left;
                    atomicAdd(iters_done, 1u);
                }
            }
        }
        __syncthreads();
        const u32 m_ex = s.bcast[1], k_ex = s.bcast[2];
#pragma unroll
        for (int q = 0; q < PL_CPT; ++q) {
            if (keep[q]) {
                u32 id = cid[q];
                if (mrg[q]) {
                    id = C - 2 - (m_ex + (ex >> 16));                                                // :311
                    node_store_plain(nodes + id, cid[q], pid[q], cb[q]);
                }
                if (PLOC_ABL != 3) entry_store(list_out, (size_t)(k_ex + (ex & 0xFFFFu)), id, cb[q]);           // :355-361
            }
            ex += ((u32)mrg[q] << 16) + (u32)keep[q];
        }
    };
    const bool stat = spec && G >= chunks;                 // (grid-uniform)
    while (true) {
        u32 chunk;
        if (stat) chunk = wg;                             // (one trip: the one-shot rule below ends the loop)
        else {
            __syncthreads();
            if (tid == 0) s.bcast[0] = atomicAdd(tickets, 1u);
            __syncthreads();
            chunk = s.bcast[0];
        }
        if (chunk >= chunks) break;
        const long long o = (long long)chunk * PLOC_CHUNK;
        // span entry k <-> list position o - HALO + k   (:232-249)
        {   // All of a thread's span entries are requested before the first is waited for (clamped positions, no branch around the loads): the loop over the span — two
            // trips for the 1024-thread shape, the second one for the 64 halo entries only — came out as load, wait, load, wait: a second dependent memory round trip
            // in every iteration's ~10 us, spent by one wave while fifteen wait at the barrier (round 4, found in the ISA).
            u32 id_[SPT]; Box b_[SPT]; bool in_[SPT]; u32 prim_[SPT];
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int k = tid + q * PL_BLOCK;
                const long long gpos = o - PL_HALO + k;
                in_[q] = k < PL_SPAN && gpos >= 0 && gpos < (long long)C;
                const size_t gc = in_[q] ? (size_t)gpos : (size_t)o;                 // (o < C: the chunk exists)
                if (FIRST) prim_[q] = svals[gc];
                else if (stat) { id_[q] = sid_[q]; b_[q] = sb_[q]; }                   // requested at the top, beside the count
                else entry_load(list_in, gc, id_[q], b_[q]);
            }
            if (FIRST) {
#pragma unroll
                for (int q = 0; q < SPT; ++q) {
                    const int k = tid + q * PL_BLOCK;
                    b_[q] = box_gather(boxes + prim_[q]); id_[q] = (u32)(in_[q] ? (size_t)(o - PL_HALO + k) : (size_t)o) + ni;
                }
            }
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const int k = tid + q * PL_BLOCK;
                if (k < PL_SPAN) {
                    if (in_[q]) {
                        lds_set(s, k, id_[q], b_[q]);
                        if (FIRST && k >= PL_HALO && k < PL_HALO + PLOC_CHUNK) {      // own = the position belongs to this chunk (not its halo): write the PrimRef
                            float* f = reinterpret_cast<float*>(leaves + (size_t)(o - PL_HALO + k));
                            reinterpret_cast<u32*>(f)[0] = prim_[q];
                            f[1] = b_[q].lx; f[2] = b_[q].ly; f[3] = b_[q].lz; f[4] = b_[q].hx; f[5] = b_[q].hy; f[6] = b_[q].hz;
                        }
                    } else s.id[k] = INV;
                    s.nn[k] = ~0ull;
                }
            }
        }
        __syncthreads();
        const int lo = (int)(o - PL_HALO < 0 ? PL_HALO - o : 0);                                   // first valid span entry
        const int hi = (int)((long long)C - (o - PL_HALO) < PL_SPAN ? (long long)C - (o - PL_HALO) : PL_SPAN);   // one past the last valid
        nn_pairs(lo, hi);
        __syncthreads();
        u32 cid[PL_CPT]; Box cb[PL_CPT]; bool mrg[PL_CPT], keep[PL_CPT]; u32 pid[PL_CPT];
        u32 packed = 0;
#pragma unroll
        for (int q = 0; q < PL_CPT; ++q) {
            const int k = PL_HALO + tid * PL_CPT + q;
            mrg[q] = false; keep[q] = false; cid[q] = INV; pid[q] = INV; cb[q] = box_empty();
            if (k < hi) {                                                                            // :274 gIdx < nClusters
                const u32 nb = (u32)s.nn[k];
                const bool mutual = (u32)s.nn[nb] == (u32)k;                                         // :276-287
                mrg[q] = mutual && (u32)k < nb; keep[q] = !mutual || mrg[q];
                cid[q] = s.id[k]; cb[q] = lds_box(s, k);
                if (mrg[q]) { pid[q] = s.id[nb]; cb[q] = box_union(cb[q], lds_box(s, (int)nb)); }
                packed += ((u32)mrg[q] << 16) + (u32)keep[q];
            }
        }
        u32 tot; const u32 ex = block_scan<PL_BLOCK>(s, packed, &tot);
        if (tid == 0) {                                // the chunk's own totals are out before anything waits
            const u64 mine = ((u64)(tot >> 16) << 31) | (u64)(tot & 0xFFFFu);
            __hip_atomic_store(status + chunk, ((chunk == 0 || PLOC_ABL == 1) ? PS_INCL : PS_LOCAL) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p_have) { __syncthreads(); finish(p_chunk, p_tot, p_ex, p_cid, p_pid, p_cb, p_mrg, p_keep); }
        p_have = true; p_chunk = chunk; p_tot = tot; p_ex = ex;
#pragma unroll
        for (int q = 0; q < PL_CPT; ++q) { p_cid[q] = cid[q]; p_pid[q] = pid[q]; p_cb[q] = cb[q]; p_mrg[q] = mrg[q]; p_keep[q] = keep[q]; }
        // the grid covers the iteration's chunks (every late iteration): when no workgroup takes a second ticket the first `chunks` tickets go to
        // `chunks` different workgroups, so nobody needs to ask again just to learn that the list is used up — the finish below starts a round trip earlier
        // Measured on the MI355X (whole build, same box): Sponza-like 262 144 0.4107 -> 0.4065 ms, 524 288 0.5216 -> 0.5180, uniform 1 M 0.5630 -> 0.5605; but 2 M
        // 0.783 -> 0.790 and 10 M 2.206 -> 2.245 (there the second ticket's round trip is what gives the predecessors time to publish before the walk): small inputs only
        if (stat || (G >= chunks && ni < (u32)PLOC_ONE_SHOT_MAX_N)) break;  // (grid-uniform; a static chunk id is good for exactly one trip)
    }
    if (p_have) { __syncthreads(); finish(p_chunk, p_tot, p_ex, p_cid, p_pid, p_cb, p_mrg, p_keep); }
    return false;
}

template <int PL_BLOCK, bool FIRST>
__global__ __launch_bounds__(PL_BLOCK, (PL_BLOCK == 512 ? PLOC_OCC : 4)) void k_ploc_iter(const float4* __restrict__ list_in, float4* __restrict__ list_out,
                                                        bvh2_node* __restrict__ nodes,
                                                        u64* status, u32* counts, u32* tickets, u32* iters_done, u32 ni,
                                                        const bvh_aabb* __restrict__ boxes, const u32* __restrict__ svals, bvh_primref* __restrict__ leaves, int static_ok) {
    __shared__ PlocLds s;
    (void)ploc_iter_body<PL_BLOCK, FIRST>(s, list_in, list_out, nodes, status, counts, tickets, iters_done, ni, boxes, svals, leaves, bid_x(), nbid_x(), static_ok != 0);
}

// one launch clears the per-iteration bookkeeping (two memsets + a one-thread kernel before: three launch boundaries of ~2 us in front of every build)
__global__ __launch_bounds__(256) void k_ploc_init(u32* __restrict__ state, u32 count, uint4* __restrict__ status, u32 status_vecs, u64* __restrict__ status_tail, u32 tail_words) {
    const u32 t = bid_x() * 256 + tid_x(), stride = nbid_x() * 256;
    for (u32 i = t; i < status_vecs; i += stride) status[i] = make_uint4(0u, 0u, 0u, 0u);
    if (t < tail_words) status_tail[t] = 0ull;
    if (t < (u32)PLOC_STATE_WORDS) state[t] = t == 0u ? count : 0u;
}

// state words: counts[0..MAX_ITERS] | tickets[0..MAX_ITERS) | iterations done
void ploc_begin(hipStream_t s, const PlocScratch& sc, uint32_t n) { ploc_reset(s, sc, n, n); }
void ploc_reset(hipStream_t s, const PlocScratch& sc, uint32_t n, uint32_t count) {
    static_assert(PLOC_STATE_WORDS <= 256, "k_ploc_init: the state words are written by the first workgroup");
    const size_t words = (size_t)PLOC_MAX_ITERS * ploc_chunks(n);          // u64 status words (the array is 256-byte aligned: 16-byte stores)
    const u32 vecs = (u32)(words / 2), tail = (u32)(words % 2);
    u32 blocks = (vecs + 255u) / 256u; if (blocks < 1u) blocks = 1u; if (blocks > 1024u) blocks = 1024u;
    hipLaunchKernelGGL(k_ploc_init, dim3(blocks), dim3(256), 0, s, sc.state, count, reinterpret_cast<uint4*>(sc.status), vecs, sc.status + (size_t)vecs * 2, tail);
}
void ploc_begin_prep(const PlocScratch& sc, uint32_t n, PrepArgs& prep) {
    const size_t words = (size_t)PLOC_MAX_ITERS * ploc_chunks(n);
    const u32 vecs = (u32)(words / 2), tail = (u32)(words % 2);
    prep.ploc_status = reinterpret_cast<uint4*>(sc.status); prep.ploc_status_vecs = vecs;
    prep.ploc_tail = sc.status + (size_t)vecs * 2; prep.ploc_tail_words = tail;
    prep.ploc_state = sc.state; prep.ploc_count = n;
}
// enqueue iterations [first, first+count) of the current batch; parity = which id buffer iteration `first` reads.  The host does not
// know the cluster count of an iteration; it only shapes the launch from a guess (C shrinks by ~20 % per iteration): any grid
// is correct because chunks are taken from a ticket counter, a good guess is merely faster.
// fresh: iteration `first` is the build's very first one (reads svals / boxes, writes the leaves: fused SetupClusters)
void ploc_enqueue(hipStream_t s, const PlocScratch& sc, uint32_t n, void* d_nodes, void* d_leaves, const void* d_boxes, const uint32_t* d_svals,
                  int first, int count, int parity, bool fresh) {
    const u32 chunks = ploc_chunks(n);
    u32* counts = sc.state; u32* tickets = sc.state + PLOC_MAX_ITERS + 1; u32* done = sc.state + 2 * PLOC_MAX_ITERS + 1;
    KernelScope ks(s, "k_ploc_iter");                   // the batch of launches is timed as one group
    for (int k = first; k < first + count; ++k) {
        const bool even = ((k + parity) & 1) == 0;
        double guess = (double)chunks; for (int j = 0; j < k; ++j) guess *= 0.83;      // chunks expected at iteration k, generously
        const bool wide = guess <= 1024.0;                                              // a few workgroups per CU at most: latency matters
        u32 grid = (u32)(2.0 * guess) + 8u; if (grid > (wide ? 512u : 1024u)) grid = wide ? 512u : 1024u; if (grid > chunks) grid = chunks;
        const float4* in = (const float4*)(even ? sc.list0 : sc.list1); float4* out = (float4*)(even ? sc.list1 : sc.list0);
        const bool f0 = fresh && k == first;
#define PLOC_LAUNCH(BLK, FIRST) hipLaunchKernelGGL((k_ploc_iter<BLK, FIRST>), dim3(grid), dim3(BLK), 0, s, in, out, (bvh2_node*)d_nodes, sc.status + (size_t)k * chunks, \
                                                   counts + k, tickets + k, done, n - 1, (const bvh_aabb*)d_boxes, d_svals, (bvh_primref*)d_leaves, sc.static_ids ? 1 : 0)
        if (wide) { if (f0) PLOC_LAUNCH(1024, true); else PLOC_LAUNCH(1024, false); }
        else      { if (f0) PLOC_LAUNCH(PLOC_NARROW, true); else PLOC_LAUNCH(PLOC_NARROW, false); }
#undef PLOC_LAUNCH
    }
}

// (kernels.hpp: touching one kernel of this translation unit makes the runtime load its code object — bvh_ctx_create does that for the build path's modules, so
// that a context's FIRST build does not pay for it: 0.3-0.7 ms per module on the MI355X, tools/cold_probe.py)
void warm_ploc() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_ploc_init)); }

} // namespace bvh
