// sort.hip — stage S: one-sweep LSD radix sort of (key, u32 value) pairs for gfx950; keys are u32 (the reference's 30-bit
// Morton codes) or u64 (60-bit codes).
//
// Replaces Oro::RadixSort::sort(KeyValueSoA src, KeyValueSoA dst, n, startBit, endBit, stream) — the reference's only
// use of the (un-vendored) Orochi library on this path; call sites src/TwoPassLbvh.cpp:71-89, src/SinglePassLbvh.cpp:72-90,
// src/PLOC++Bvh.cpp:62-80, src/Hploc.cpp:63-81.  Contract: stable ascending order on key bits [start,end).
//
// Design (CDNA4):
//  * 8-bit digits; every pass reads each pair once and writes it once (16 B per pair per pass) — the digit histograms of
//    ALL passes are produced up front (fused into the Morton kernel, or by k_hist for the stand-alone entry point), so no
//    pass re-reads keys to count; every tile scans the 256 raw counts of its pass itself (no scan launch).
//  * one workgroup sorts a tile: 512 threads x 13 pairs = 6656 pairs from SORT_WIDE_MIN_N keys on (two workgroups per CU, 61 KB of LDS), 1024 threads x 3
//    pairs = 3072 below (a pass is then one generation of tiles: what counts is a tile's latency).  Per-wave ranking by ballot "match-any" (one ballot
//    per digit bit and key, no LDS atomics, stable by construction), per-wave digit counters in LDS, a cross-wave scan, then the tile's digit totals are
//    chained to earlier tiles by decoupled look-back on 32-bit status words {flag:2, count:30}.  Status words are relaxed agent-scope atomics: the value
//    is its own flag, so no fence is needed across XCDs.
//  * tile id = workgroup id (no ticket: a returning atomic on one word per tile cost 10 us of every pass).  Progress under ANY dispatch order comes from
//    helping: a thread that has polled an unpublished predecessor SORT_HELP_AFTER times counts that tile's keys for its digit itself and publishes the
//    total on its behalf (idempotent).
//  * the digit width is a template parameter (round 4): the build sorts its 30-bit Morton codes as 8 / 8 / 8 / 6 bits, and the 6-bit last pass runs with 64
//    digit threads, 6 ballots per key and 64 status words per tile instead of 256 (192 of them were structurally zero).
//  * pairs are staged through LDS in their tile-sorted order so that global writes are runs of consecutive addresses.
#include <cstdlib>
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

static_assert(SORT_BLOCK == SORT_RADIX, "one thread per digit (k_onesweep: the first 2^BITS threads of the workgroup)");
constexpr u32 ST_LOCAL = 1u << 30, ST_INCL = 2u << 30, ST_MASK = (1u << 30) - 1u;

// interleaved {key, value} records of the intermediate passes: 8 bytes for u32 keys, 16 bytes {key, value, pad} for u64 keys
template <typename T> __device__ __forceinline__ T sort_ld(const T* p) { return *p; }
template <typename T> __device__ __forceinline__ void sort_st(T* p, T v) { *p = v; }
#define SORT_LD(P) sort_ld(P)
#define SORT_ST(P, V) sort_st(P, V)
template <typename K> struct PairRec;
template <> struct PairRec<u32> {
    using type = u64;
    static __device__ __forceinline__ type pack(u32 k, u32 v) { return (u64)k | ((u64)v << 32); }
    static __device__ __forceinline__ type pad() { return 0xFFFFFFFFull; }                 // key all ones, value 0
    static __device__ __forceinline__ u32 key(type r) { return (u32)r; }
    static __device__ __forceinline__ u32 val(type r) { return (u32)(r >> 32); }
};
template <> struct PairRec<u64> {
    using type = uint4;
    static __device__ __forceinline__ type pack(u64 k, u32 v) { return make_uint4((u32)k, (u32)(k >> 32), v, 0u); }
    static __device__ __forceinline__ type pad() { return make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u); }
    static __device__ __forceinline__ u64 key(type r) { return (u64)r.x | ((u64)r.y << 32); }
    static __device__ __forceinline__ u32 val(type r) { return r.z; }
};

// stand-alone histogram (all passes in one read of the keys)
template <typename K>
__global__ __launch_bounds__(SORT_BLOCK) void k_hist(const K* __restrict__ keys, u32 n, int start_bit, int end_bit, int passes,
                                                     u32* __restrict__ hist) {
    __shared__ u32 s_hist[SORT_MAX_PASSES * SORT_RADIX];
    __shared__ u32 s_pad[SORT_HIST_GROUP >= 3 ? 64 : 1];
    for (int i = threadIdx.x; i < passes * SORT_RADIX; i += SORT_BLOCK) s_hist[i] = 0;
    __syncthreads();
    const u32 stride = gridDim.x * SORT_BLOCK;
    for (u32 i = blockIdx.x * SORT_BLOCK + threadIdx.x; i < n; i += stride) {
        const K k = keys[i];
        hist_add_passes<SORT_HIST_GROUP>(s_hist, passes, SORT_RADIX, [&](int p) {
            const int sh = start_bit + p * SORT_BITS;
            const int w = min(SORT_BITS, end_bit - sh);
            return (u32)(k >> sh) & ((1u << w) - 1u);
        }, s_pad);
    }
    __syncthreads();
    u32* copy = hist + (blockIdx.x % SORT_HIST_COPIES) * SORT_HIST_STRIDE;
    for (int i = threadIdx.x; i < passes * SORT_RADIX; i += SORT_BLOCK) { const u32 c = s_hist[i]; if (c) atomicAdd(&copy[i], c); }
}

// IN_AOS / OUT_AOS: the pair arrays of the intermediate passes are interleaved {key, value} u64 words — a tile's run for one
// digit is then 16 x 8 B = a full 128-byte line instead of two 64-byte half lines (measured: scattered SoA writes cost 36 % of a
// pass).  The caller-facing arrays of the first and last pass stay SoA (KeyValueSoA of Oro::RadixSort::sort).
#ifndef SORT_HELP_AFTER
#define SORT_HELP_AFTER 4096u     // empty polls (~1.7 us each) before a thread computes an unpublished predecessor's total itself
#endif
// The large-input variant (n >= SORT_WIDE_MIN_N).  u32 keys: 512 threads x 13 keys = 6656-pair tiles, 61 KB of LDS, two workgroups = 16 waves per CU.
// Same box, four passes of the stand-alone sort at 10 M: 256 x 20 (round 1 / early round 2) 0.266 ms, 512 x 10 (same tile, twice the waves) 0.265,
// 512 x 12 / 13 / 14: 0.245 / 0.241 / 0.245, 512 x 16 (one workgroup per CU) 0.298, 768 x 8: 0.305, 1024 x 8: 0.272 (but 0.074 vs 0.079 at 1 M) —
// what pays is fewer, larger tiles (fewer status rows for everybody's look-back) as long as two workgroups still fit a CU.
// Inputs below SORT_WIDE_MIN_N: a pass is one generation of tiles (262 k keys = 85 tiles on 256 CUs), so what counts is a tile's latency: the same
// 3072-pair tile on 1024 threads x 3 keys instead of 256 x 12 (four passes of the stand-alone sort, 256 x 12 -> 512 x 6 -> 1024 x 3: 50 k 0.040 -> 0.035 -> 0.034 ms,
// 262 k 0.054 -> 0.050 -> 0.049, 900 k 0.078 -> 0.073 -> 0.072).
constexpr int SORT_NARROW_NT = 1024, SORT_NARROW_IPT = SORT_TILE / 1024;
static_assert(SORT_NARROW_NT * SORT_NARROW_IPT == SORT_TILE, "the status rows are sized for SORT_TILE-pair tiles");
#ifndef SORT_WIDE_IPT
#define SORT_WIDE_IPT 13
#endif
template <typename K> struct SortWide { static constexpr int NT = 512, IPT = SORT_WIDE_IPT; };
template <> struct SortWide<u64> { static constexpr int NT = 512, IPT = 10; };    // 5120-pair tiles, 68 KB: 8 passes at 10 M 0.656 (256 x 20) -> 0.610 ms; 512 x 8: 0.657, 512 x 12: 0.731
// BITS: digit width of this instantiation's pass (6..8; a pass whose digit is narrower than BITS passes a smaller digit_mask).  Status rows keep their
// SORT_RADIX-word stride; a pass only touches the first 2^BITS words of a row.
// The tile's LDS (the kernel declares it once: the gated last pass holds two bodies, see k_onesweep)
template <typename K, int TILE, int NW> struct SortLds {
    u32 whist[NW][SORT_RADIX];       // per-wave digit counters (a body of BITS < 8 uses the first 2^BITS words of a row)
    u32 binoff[SORT_RADIX];
    u32 gbase[SORT_RADIX];
    K keys[TILE];
    u32 vals[TILE];
    u64 wsum[NW];
#ifdef BVH_ABLATION
    u32 tile;                        // (ticket order of the measurement build)
#endif
};
template <typename K, bool IOTA, bool IN_AOS, bool OUT_AOS, int IPT, int NT, int BITS>
__device__ __forceinline__ void onesweep_tile(SortLds<K, NT * IPT, NT / WAVE>& lds, const K* __restrict__ keys_in, const u32* __restrict__ vals_in,
                                              K* __restrict__ keys_out, u32* __restrict__ vals_out, u32 n,
                                              int shift, u32 digit_mask, const u32* __restrict__ ghist,
                                              u32* status, u32* tile_counter, int dbg) {
    constexpr int RADIX = 1 << BITS;
    constexpr int NW = NT / WAVE, NDW = RADIX / WAVE;             // waves; waves that own digits (threads 0 .. RADIX-1: one digit each)
    constexpr int TILE = NT * IPT;                   // keys per workgroup
    static_assert(BITS >= 6 && BITS <= SORT_BITS && NT % RADIX == 0, "digit threads are whole waves");
    u32 (&s_whist)[NW][SORT_RADIX] = lds.whist;
    u32 (&s_binoff)[SORT_RADIX] = lds.binoff;
    u32 (&s_gbase)[SORT_RADIX] = lds.gbase;
    using Rec = PairRec<K>;
    K (&s_keys)[TILE] = lds.keys;
    u32 (&s_vals)[TILE] = lds.vals;
    u64 (&s_wsum)[NW] = lds.wsum;
#ifdef BVH_ABLATION
    u32& s_tile = lds.tile;
#endif

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
#ifdef BVH_ABLATION
    u64 ts[8]; int nts = 0;
#define SORT_STAMP() do { if (dbg & 16) ts[nts] = __builtin_amdgcn_s_memtime(); ++nts; } while (0)
#else
#define SORT_STAMP() do { } while (0)
#endif
    SORT_STAMP();                                    // 0: start
    // Tile id = workgroup id.  Decoupled look-back makes a tile wait for its predecessors' totals; with static ids that is only deadlock-free
    // if every predecessor is (or gets) resident, which HIP's dispatch order does not promise.  The usual cure — tile ids from an atomic
    // ticket — costs a returning atomic on ONE word per tile (~90 per us on this chip): the ~1000 workgroups that start together queued up to
    // 11 us for it, 10 us of every 68-us pass at 10 M.  Here progress is guaranteed differently: a thread that has polled an unpublished
    // predecessor SORT_HELP_AFTER times counts that tile's keys for its digit itself and publishes the total on the predecessor's behalf
    // (idempotent: the owner would write the same number), so every resident tile finishes in bounded time whatever else is scheduled.  In
    // order dispatch never takes that path; BVH_SORT_DEBUG=8 reverses the tile order and 32 helps at the first empty poll, which is how
    // tests/test_gpu_round2.py exercises it.
#ifdef BVH_ABLATION
    if (tid == 0) s_tile = (dbg & 64) ? atomicAdd(tile_counter, 1u) : (dbg & 8) ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
#endif
    const bool dig = tid < RADIX;                    // this thread speaks for digit `tid`
    // every wave clears ITS row of the per-wave digit counters (the ranking only touches the wave's own row, and a wave's LDS operations execute in order): no barrier
    // between the clearing and the ranking, and the tile's key loads go out at once (round 4: the tile id used to come through LDS behind a barrier)
    for (int d = lane; d < RADIX; d += WAVE) s_whist[wave][d] = 0;
#ifdef BVH_ABLATION
    __syncthreads();
    const u32 tile = s_tile;
#else
    const u32 tile = (dbg & 8) ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
#endif
    const u32 base = tile * (u32)TILE;
    const u32 valid = min((u32)TILE, n - base);

    // ---- load (wave-striped: wave w owns a contiguous 64*IPT span, item i is a coalesced 256-B row of it)
    K key[IPT]; u32 val[IPT], pos[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const u32 local = (u32)(wave * WAVE * IPT + i * WAVE + lane);
        const bool ok = local < valid;
        if (IN_AOS) {
            const typename Rec::type kv = ok ? SORT_LD(reinterpret_cast<const typename Rec::type*>(keys_in) + base + local) : Rec::pad();   // (a select, not a
            key[i] = Rec::key(kv); val[i] = Rec::val(kv);                                   //  branch: the 16 loads stay in flight together)
        } else {
            key[i] = ok ? SORT_LD(keys_in + base + local) : ~(K)0;
            val[i] = IOTA ? (base + local) : (ok ? SORT_LD(vals_in + base + local) : 0u);
        }
    }
#ifdef BVH_ABLATION
    if (dbg & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SORT_STAMP();                                    // 1: ticket + loads landed
    // ---- rank inside the wave: lanes holding the same digit form a group (8 ballots); every member reads the wave's LDS
    // counter for that digit, the group's lowest lane bumps it; rank = counter-before + index inside the group.  Program order
    // (item-major, then lane) is exactly memory order inside the wave's span => stable.
    const u64 lt = lanemask_lt();
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const u32 d = (u32)(key[i] >> shift) & digit_mask;
        // lanes whose digit differs from mine in bit b: ballot(bit b) xor (my bit b, sign-extended); the group is what is left
        u32 diff_lo = 0u, diff_hi = 0u;
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const int mine = __builtin_amdgcn_sbfe((int)d, b, 1);           // 0 or -1
            const u64 bal = __ballot(mine != 0);
            diff_lo |= (u32)bal ^ (u32)mine; diff_hi |= (u32)(bal >> 32) ^ (u32)mine;
        }
        const u64 grp = ~(((u64)diff_hi << 32) | diff_lo);
        const u32 below = (u32)__popcll(grp & lt);
        // every member reads the counter (same address -> LDS broadcast), then the group's lowest lane bumps it; DS
        // operations of a wave execute in order, so the next item's read sees the update without a cross-lane hop
        const u32 before = s_whist[wave][d];
        if (below == 0) s_whist[wave][d] = before + (u32)__popcll(grp);
        pos[i] = before + below;
    }
    __syncthreads();
    SORT_STAMP();                                    // 2: ranked

    // ---- digit `tid`: totals over the 4 waves, exclusive wave offsets back into s_whist, publish the tile aggregate
    u32 total = 0;
    if (dig) {
        u32 run = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const u32 c = s_whist[w][tid]; s_whist[w][tid] = run; run += c; }
        total = run;
        if ((u32)tid == digit_mask) total -= (u32)TILE - valid;     // padding keys (all ones) carry the top digit; they are not data
        st_agent(&status[(size_t)tile * SORT_RADIX + tid], (tile == 0 ? ST_INCL : ST_LOCAL) | total);
    }
    // ---- exclusive scans over the 256 digits (wave scan + LDS hop), two in one: the tile's digit totals -> s_binoff, and the pass's
    // raw global digit counts -> gexcl (every tile redoes that 256-entry scan from L2: cheaper than a kernel launch per sort)
    u32 gexcl = 0;
    {
        u32 graw = 0;
        u64 inc = 0;
        if (dig) {                                   // (whole waves)
#pragma unroll
            for (int c = 0; c < SORT_HIST_COPIES; ++c) graw += ghist[c * SORT_HIST_STRIDE + tid];  // (independent loads, L2 hits)
            inc = ((u64)graw << 32) | total;
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) { const u64 t = __shfl_up(inc, off); if (lane >= off) inc += t; }
            if (lane == WAVE - 1) s_wsum[wave] = inc;
        }
        __syncthreads();
        if (dig) {
            u64 wbase = 0;
#pragma unroll
            for (int w = 0; w < NDW; ++w) if (w < wave) wbase += s_wsum[w];
            const u64 ex = wbase + inc - (((u64)graw << 32) | total);
            s_binoff[tid] = (u32)ex; gexcl = (u32)(ex >> 32);
        }
    }
    // ---- tile-local sort through LDS, ahead of the look-back: it needs nothing from other tiles, and the predecessors publish meanwhile
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const u32 d = (u32)(key[i] >> shift) & digit_mask;
        const u32 p = s_binoff[d] + s_whist[wave][d] + pos[i];
        s_keys[p] = key[i]; s_vals[p] = val[i];
    }
    SORT_STAMP();                                    // 3: totals published, scans, exchange writes
    // ---- decoupled look-back for digit `tid`: LB_WINDOW predecessors are fetched per step (independent loads in flight)
    // so that a walk over k tiles costs ~k/LB_WINDOW memory round trips instead of k.  Measured at 10 M (BVH_SORT_DEBUG=16 in the
    // ablation build): 9.4 steps per tile, 2.8 of them empty polls; skipping the look-back altogether takes a pass from 70 to
    // 45 us, but neither wider windows (16: 74 us, 32: 83 us), nor 8-tile skip words (5.5 steps: 68 us), nor more workgroups per CU,
    // nor dedicated scanner waves (tiles polling one word: 230+ us, the scanners' round trips serialise) recover any of it: what
    // costs is that every tile keeps its CU slot until its slowest predecessor has published.
    {
        u32 excl = 0;
        if (dig && tile > 0 && !(dbg & 1)) {      // (dbg & 1, & 2: ablation build only)
            constexpr int LB_WINDOW = 8;
            int prev = (int)tile - 1;
            bool done = false;
            u32 stalled = 0;
            const u32 help_after = (dbg & 32) ? 1u : SORT_HELP_AFTER;
#ifdef BVH_ABLATION
            u32 n_steps = 0, n_empty = 0;
#endif
            while (!done) {
                u32 st[LB_WINDOW];
#pragma unroll
                for (int w = 0; w < LB_WINDOW; ++w)
                    st[w] = (prev - w >= 0) ? ld_agent(&status[(size_t)(prev - w) * SORT_RADIX + tid]) : (ST_INCL | 0u);
                int used = 0;
#pragma unroll
                for (int w = 0; w < LB_WINDOW; ++w) {
                    if (done || used != w) continue;
                    const u32 flag = st[w] >> 30;
                    if (flag == 0) continue;                 // not published yet: re-poll from here
                    excl += st[w] & ST_MASK; used = w + 1;
                    if (flag == 2) done = true;
                }
                prev -= used;
#ifdef BVH_ABLATION
                ++n_steps; if (used == 0) ++n_empty;
#endif
                if (!done && used == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++stalled >= help_after) {           // predecessor `prev` may never get to run: publish its total for digit `tid` for it
                        const u32 pb = (u32)prev * (u32)TILE, pv = min((u32)TILE, n - pb);
                        u32 c = 0;
                        for (u32 i = 0; i < pv; ++i) {
                            const K k = IN_AOS ? Rec::key(reinterpret_cast<const typename Rec::type*>(keys_in)[pb + i]) : keys_in[pb + i];
                            c += ((u32)(k >> shift) & digit_mask) == (u32)tid ? 1u : 0u;
                        }
                        st_agent(&status[(size_t)prev * SORT_RADIX + tid], (prev == 0 ? ST_INCL : ST_LOCAL) | c);
                        stalled = 0;
                    }
                } else stalled = 0;
            }
#ifdef BVH_ABLATION
            if ((dbg & 16) && tid == 0) { atomicAdd(tile_counter + 4, n_steps); atomicAdd(tile_counter + 8, n_empty); atomicMax(tile_counter + 12, n_steps); }
#endif
            st_agent(&status[(size_t)tile * SORT_RADIX + tid], ST_INCL | (excl + total));
        }
#ifdef BVH_ABLATION
        if (dig && tile > 0 && (dbg & 1)) excl = tile * total;     // timing only: the tile pretends its predecessors held what it holds, so that the scatter keeps a realistic address pattern
#endif
        if (dig) s_gbase[tid] = gexcl + excl - s_binoff[tid];
    }
    SORT_STAMP();                                    // 4: own look-back done
    __syncthreads();
    SORT_STAMP();                                    // 5: everybody's look-back done

#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        const u32 p = (u32)(k * NT + tid);
        if (p < valid) {
            const K kk = s_keys[p];
#ifdef BVH_ABLATION
            u32 dst = (dbg & 2) ? base + p : s_gbase[(u32)(kk >> shift) & digit_mask] + p;
            if ((dbg & 1) && dst >= n) dst = n - 1u;                // (the pretended offsets can run a little past the end)
#else
            const u32 dst = s_gbase[(u32)(kk >> shift) & digit_mask] + p;
#endif
            if (OUT_AOS) SORT_ST(reinterpret_cast<typename Rec::type*>(keys_out) + dst, Rec::pack(kk, s_vals[p]));
            else { SORT_ST(keys_out + dst, kk); SORT_ST(vals_out + dst, s_vals[p]); }
        }
    }
#ifdef BVH_ABLATION
    if (dbg & 16) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SORT_STAMP();                                // 6: scattered
        if (tid == 0) for (int k = 1; k < 7; ++k) atomicAdd(tile_counter + 16 + k, (u32)(ts[k] - ts[k - 1]));
    }
#endif
#undef SORT_STAMP
}
// GATED (the build's last u32 pass from SORT_WIDE_MIN_N keys on; SORT_WIDE_FLAG_WORD, kernels.hpp): ONE launch holds both bodies of the pass — the narrow [24, 30) one
// (64 digit threads, 6 ballots per key) and the reference's [24, 32) one — and every workgroup takes the branch the word k_morton raised selects: a scalar load and a
// scalar branch in front of the tile's loads.  (Round 5 enqueued two launches, each leaving at once when it was not its turn: + 4.8 us per build for the second one,
// which is what the narrow pass saves.)  ghist + SORT_RADIX is that word for the fourth pass; BITS / digit_mask are the narrow body's.
template <typename K, bool IOTA, bool IN_AOS, bool OUT_AOS, int IPT, int NT = SORT_BLOCK, int BITS = SORT_BITS, bool GATED = false>
__global__ __launch_bounds__(NT) void k_onesweep(const K* __restrict__ keys_in, const u32* __restrict__ vals_in,
                                                         K* __restrict__ keys_out, u32* __restrict__ vals_out, u32 n,
                                                         int shift, u32 digit_mask, const u32* __restrict__ ghist,
                                                         u32* status, u32* tile_counter, int dbg) {
    __shared__ SortLds<K, NT * IPT, NT / WAVE> lds;
    if (GATED && ghist[SORT_RADIX] != 0u)
        onesweep_tile<K, IOTA, IN_AOS, OUT_AOS, IPT, NT, SORT_BITS>(lds, keys_in, vals_in, keys_out, vals_out, n, shift, (u32)SORT_RADIX - 1u, ghist, status, tile_counter, dbg);
    else
        onesweep_tile<K, IOTA, IN_AOS, OUT_AOS, IPT, NT, BITS>(lds, keys_in, vals_in, keys_out, vals_out, n, shift, digit_mask, ghist, status, tile_counter, dbg);
}

#ifdef BVH_ABLATION
__global__ void k_lb_report(u32* c, u32 tiles, int pass) {
    printf("pass %d: %u tiles, look-back steps/tile %.2f (max %u), empty polls/tile %.2f\n", pass, tiles, (double)c[4] / tiles, c[12], (double)c[8] / tiles);
    printf("   thread 0's clock ticks per tile: load %.0f  rank %.0f  totals+scan+exchange %.0f  look-back %.0f  wait for the block %.0f  scatter %.0f\n",
           (double)c[17] / tiles, (double)c[18] / tiles, (double)c[19] / tiles, (double)c[20] / tiles, (double)c[21] / tiles, (double)c[22] / tiles);
    for (int k = 16; k < 24; ++k) c[k] = 0u;
}
#endif
__global__ void k_iota(u32* __restrict__ out, u32 n) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = i;
}

size_t sort_status_bytes(uint32_t n) { return (size_t)SORT_MAX_PASSES * sort_tiles(n) * SORT_RADIX * sizeof(u32); }

// One launch clears everything a build needs cleared: the digit histograms, the look-back status words and tile tickets of `passes`
// sort passes, optionally the scene extent (Aabb::reset) and one more small word array (the emitters' queue heads).  (Four memsets
// and a reset kernel cost ~4 us each of launch latency: a fifth of a 262 k build.)
__global__ __launch_bounds__(256) void k_prepare(u32* __restrict__ hist, u32 hist_words, uint4* __restrict__ status, u32 status_vecs,
                                                 u32* __restrict__ counters, float* __restrict__ scene, u32* __restrict__ extra, u32 extra_words) {
    const u32 t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (u32 i = t; i < status_vecs; i += stride) status[i] = make_uint4(0u, 0u, 0u, 0u);
    for (u32 i = t; i < hist_words; i += stride) hist[i] = 0u;
    for (u32 i = t; i < extra_words; i += stride) extra[i] = 0u;
#ifdef BVH_ABLATION
    if (t < 64u) counters[t] = 0u;                   // (+ the look-back statistics of the measurement build; the array is padded to 256 bytes)
#else
    if (t < (u32)SORT_MAX_PASSES) counters[t] = 0u;
#endif
    if (scene && t < 6u) scene[t] = t < 3u ? FMAX : -FMAX;
}

void sort_prepare(hipStream_t s, const SortScratch& sc, uint32_t n, int passes, float* d_scene_reset, uint32_t* d_extra, uint32_t extra_words) {
    const u32 hist_words = (u32)SORT_HIST_COPIES * SORT_HIST_STRIDE; (void)passes;
    const size_t status_words = (size_t)passes * sort_tiles(n) * SORT_RADIX;          // a multiple of 4 (SORT_RADIX = 256), 16-byte aligned base
    const u32 vecs = (u32)(status_words / 4);
    u32 blocks = (vecs + 255u) / 256u; if (blocks < 8u) blocks = 8u; if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(k_prepare, dim3(blocks), dim3(256), 0, s, sc.hist, hist_words, reinterpret_cast<uint4*>(sc.status), vecs, sc.counters,
                       d_scene_reset, d_extra, extra_words);
}

template <typename K>
static void sort_pairs_t(hipStream_t s, const SortScratch& sc, const K* keys_in, const uint32_t* vals_in, uint32_t n,
                         K* keys_out, uint32_t* vals_out, int start_bit, int end_bit, bool hist_ready, bool gated_narrow_top = false) {
    const int passes = sort_passes(start_bit, end_bit);
    if (passes <= 0) {   // nothing to sort on: identity permutation
        (void)hipMemcpyAsync(keys_out, keys_in, (size_t)n * sizeof(K), hipMemcpyDeviceToDevice, s);
        if (vals_in) (void)hipMemcpyAsync(vals_out, vals_in, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
        else hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, s, vals_out, n);
        return;
    }
    // small inputs: 256 threads x 12 keys (4 passes at 262 k: 16 keys 0.057 ms, 12: 0.054, 8: 0.053, 4: 0.058); from SORT_WIDE_MIN_N on: SortWide (round 1:
    // 256 x 20 instead of 256 x 16, same box, 4 passes: 10 M 0.299 -> 0.280 ms, 2 M 0.114 -> 0.113, 262 k 0.064 -> 0.067:
    // fewer tiles = fewer status rows for everybody's look-back, but a longer critical path per tile)
    const bool wide = n >= SORT_WIDE_MIN_N;
    const u32 tile_keys = wide ? (u32)SortWide<K>::NT * SortWide<K>::IPT : (u32)SORT_NARROW_NT * SORT_NARROW_IPT;
    const u32 tiles = (n + tile_keys - 1u) / tile_keys;           // (<= sort_tiles(n): the status rows were sized and cleared for that)
    if (!hist_ready) {
        const u32 blocks = (n + SORT_BLOCK - 1) / SORT_BLOCK;
        KernelScope ks(s, "k_hist");
        hipLaunchKernelGGL(k_hist<K>, dim3(blocks < 1024u ? blocks : 1024u), dim3(SORT_BLOCK), 0, s, keys_in, n, start_bit, end_bit, passes, sc.hist);
    }
#ifdef BVH_ABLATION
    static const int env_dbg = getenv("BVH_SORT_DEBUG") ? atoi(getenv("BVH_SORT_DEBUG")) : 0;   // ablation build, measurements only: results are wrong when bits 1 / 2 are set
    const int dbg = env_dbg | (sc.test_knobs & (8 | 32));          // (bit 128: the LAST pass runs without its look-back — in-bounds garbage nobody consumes; what a pass without look-back would save)
#else
    const int dbg = sc.test_knobs & (8 | 32);              // BVH_OPT_SORT_TEST_KNOBS: test knobs of the helping path (results stay right)
#endif
    const K* kin = keys_in; const u32* vin = vals_in;
    for (int p = 0; p < passes; ++p) {
        const bool first = p == 0, last = p == passes - 1;
        // intermediate pair arrays: interleaved ping-pong buffers pairs0 / pairs1
        K* kout = last ? keys_out : reinterpret_cast<K*>((p & 1) ? sc.pairs1 : sc.pairs0);
        u32* vout = last ? vals_out : nullptr;
        const int sh = start_bit + p * SORT_BITS;
        const int w = (end_bit - sh) < SORT_BITS ? (end_bit - sh) : SORT_BITS;
        const u32 mask = (1u << w) - 1u;
        u32* st = sc.status + (size_t)p * tiles * SORT_RADIX;
        const u32* h = sc.hist + p * SORT_RADIX;
        u32* tc = sc.counters + p;
        KernelScope ks(s, "k_onesweep");
        const dim3 g(tiles), bn(SORT_NARROW_NT), bw(SortWide<K>::NT);
        const int pdbg = dbg | (((dbg & 128) && last) ? 1 : 0);
#define SWEEP_G(IOTA, INA, OUTA, BB, GG, MASK) do { if (wide) hipLaunchKernelGGL((k_onesweep<K, IOTA, INA, OUTA, SortWide<K>::IPT, SortWide<K>::NT, BB, GG>), g, bw, 0, s, kin, vin, kout, vout, n, sh, MASK, h, st, tc, pdbg); \
                                  else      hipLaunchKernelGGL((k_onesweep<K, IOTA, INA, OUTA, SORT_NARROW_IPT, SORT_NARROW_NT, BB, GG>), g, bn, 0, s, kin, vin, kout, vout, n, sh, MASK, h, st, tc, pdbg); } while (0)
#define SWEEP_B(IOTA, INA, OUTA, BB) SWEEP_G(IOTA, INA, OUTA, BB, false, mask)
#define SWEEP(IOTA, INA, OUTA) SWEEP_B(IOTA, INA, OUTA, SORT_BITS)
        bool launched = false;
        if constexpr (sizeof(K) == 4) {
            if (last && !first && gated_narrow_top && w == 6 && p == 3 && start_bit == 0 && hist_ready) {
                SWEEP_G(false, true, false, 6, true, mask);               // the build's [24, 30) pass — or, when a code has bit 30 / 31 set, the reference's [24, 32) pass
                launched = true;                                          // (same launch, same rows, same histogram)
            }
        }
        if (launched) { }
        else if (first && last) { if (vin == nullptr) SWEEP(true, false, false); else SWEEP(false, false, false); }
        else if (first)         { if (vin == nullptr) SWEEP(true, false, true);  else SWEEP(false, false, true); }
        else if (last)          { if (w <= 6) SWEEP_B(false, true, false, 6); else if (w == 7) SWEEP_B(false, true, false, 7); else SWEEP(false, true, false); }   // a narrow top digit:
        else                    SWEEP(false, true, true);                                                                   // fewer digit threads, ballots and status words
#undef SWEEP_G
#undef SWEEP_B
#undef SWEEP
#ifdef BVH_ABLATION
        if (dbg & 16) hipLaunchKernelGGL(k_lb_report, dim3(1), dim3(1), 0, s, tc, tiles, p);
#endif
        kin = kout; vin = vout;
    }
}

void sort_pairs(hipStream_t s, const SortScratch& sc, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t n,
                uint32_t* keys_out, uint32_t* vals_out, int start_bit, int end_bit, bool hist_ready, bool gated_narrow_top) {
    sort_pairs_t<u32>(s, sc, keys_in, vals_in, n, keys_out, vals_out, start_bit, end_bit, hist_ready, gated_narrow_top);
}
void sort_pairs64(hipStream_t s, const SortScratch& sc, const uint64_t* keys_in, const uint32_t* vals_in, uint32_t n,
                  uint64_t* keys_out, uint32_t* vals_out, int start_bit, int end_bit, bool hist_ready) {
    sort_pairs_t<u64>(s, sc, keys_in, vals_in, n, keys_out, vals_out, start_bit, end_bit, hist_ready);
}

// (kernels.hpp: touching one kernel of this translation unit makes the runtime load its code object — bvh_ctx_create does that for the build path's modules, so
// that a context's FIRST build does not pay for it: 0.3-0.7 ms per module on the MI355X, tools/cold_probe.py)
void warm_sort() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_iota)); }

} // namespace bvh
