// common.hpp — device helpers shared by the gfx950 kernels of the BVH build path.
// Wave width is hard-coded to 64 (CDNA4).  Compiled with -ffp-contract=off: the area expression and the Morton
// quantisation must round exactly like the CPU oracle (ties on area bit patterns decide PLOC/HPLOC merges).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bvh/types.h"

namespace bvh {

using u32 = uint32_t;
using u64 = uint64_t;
constexpr u32 INV = BVH_INVALID;
constexpr float FMAX = BVH_FLT_MAX;
constexpr int WAVE = 64;

struct Box { float lx, ly, lz, hx, hy, hz; };   // = bvh_aabb / reference Aabb (24 B)

__device__ __forceinline__ Box box_empty() { return { FMAX, FMAX, FMAX, -FMAX, -FMAX, -FMAX }; }   // Aabb::reset, src/Common.h:327-331
__device__ __forceinline__ Box box_union(const Box& a, const Box& b) {                              // Aabb::grow / merge, :333-338,:456-459
    return { fminf(a.lx, b.lx), fminf(a.ly, b.ly), fminf(a.lz, b.lz), fmaxf(a.hx, b.hx), fmaxf(a.hy, b.hy), fmaxf(a.hz, b.hz) };
}
__device__ __forceinline__ float box_area(const Box& b) {                                           // Aabb::area, :361-365 (no FMA)
    const float ex = b.hx - b.lx, ey = b.hy - b.ly, ez = b.hz - b.lz;
    return 2 * (ex * ey + ex * ez + ey * ez);
}

// ---- agent-scope (device-wide, cross-XCD) accesses -------------------------------------------------------------
// MI355X has 8 XCDs with private L2s and per-CU L1s that other CUs' stores never refresh.  Data handed between
// workgroups inside one launch therefore travels as relaxed agent-scope atomics (global_load/store ... sc1: L1 bypass,
// write-through) and the hand-off word itself is an agent-scope atomic RMW issued after the producer drained its
// stores (s_waitcnt vmcnt(0)).  See DESIGN.md "Inter-workgroup hand-off".
__device__ __forceinline__ u32  ld_agent(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64  ld_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
// Work-item / workgroup ids as compiler builtins.  The emit kernels are compiled with -mno-amdgpu-ieee; threadIdx / blockIdx / gridDim go through
// __ockl_get_local_id & co., device-library functions compiled WITH the IEEE mode, which the inliner then refuses to inline: every kernel started with
// two or three real function calls (s_swappc), 64-bit ids of unknown range and a 32-VGPR floor.  The builtins carry the launch bounds' ranges.
__device__ __forceinline__ u32 tid_x() { return __builtin_amdgcn_workitem_id_x(); }
__device__ __forceinline__ u32 bid_x() { return __builtin_amdgcn_workgroup_id_x(); }
__device__ __forceinline__ u32 bdim_x() { return __builtin_amdgcn_workgroup_size_x(); }
// workgroups of the launch: hidden_block_count_x, the first word of the implicit kernel arguments (code object v5: what __ockl_get_num_groups(0) reads).  Not
// __builtin_amdgcn_grid_size_x(): that one loads from the AQL dispatch packet — host memory, a microsecond per wave (measured: PLOC++ at 262 144 0.39 -> 0.48 ms).
__device__ __forceinline__ u32 nbid_x() { return ((const u32*)__builtin_amdgcn_implicitarg_ptr())[0]; }

__device__ __forceinline__ u64 pack2(float a, float b) { return (u64)__float_as_uint(a) | ((u64)__float_as_uint(b) << 32); }
__device__ __forceinline__ float lo_f(u64 v) { return __uint_as_float((u32)v); }
__device__ __forceinline__ float hi_f(u64 v) { return __uint_as_float((u32)(v >> 32)); }

// Bvh2Node (32 B, 32-B aligned) written as two 16-byte write-through (sc1) stores: a dwordx2 sc1 store costs ~2.7x a dwordx4
// one per byte on this chip (MI355X_MICROARCH.md, stores table).  Inline asm: hipcc has no builtin for a 16-byte global store
// with the sc1 bit; the trailing s_nop keeps the data registers alive until the store has read them (guide §5.7 item 1),
// completion is awaited by drain_stores() before the publishing atomic, as for every other agent-scope store.
__device__ __forceinline__ void node_store_agent(bvh2_node* n, u32 left, u32 right, const Box& b) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f q0 = { __uint_as_float(left), __uint_as_float(right), b.lx, b.ly };
    const v4f q1 = { b.lz, b.hx, b.hy, b.hz };
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\ts_nop 1"
                 :: "v"(n), "v"(q0), "v"(q1) : "memory");
}
__device__ __forceinline__ Box node_box_agent(const bvh2_node* n) {
    // two 16-byte sc1 loads (8-byte agent-scope accesses run at 0.54-0.70x the 16-byte rate); the wait is part of the
    // statement because hipcc does not count loads issued from inline asm (guide §5.7 item 1, form (i))
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f q0, q1;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1) : "v"(n) : "memory");
    return { q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
}
// a whole 32-byte record {w0, w1, box} (Bvh2Node layout) written by node_store_agent, possibly by another workgroup of this launch
__device__ __forceinline__ void rec_load_agent(const bvh2_node* n, u32& w0, u32& w1, Box& b) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f q0, q1;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(q0), "=&v"(q1) : "v"(n) : "memory");
    w0 = __float_as_uint(q0.x); w1 = __float_as_uint(q0.y);
    b = { q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
}
// the same load in two steps — request now, wait later — so that other loads can be in flight beside it (k_hploc_ext's work list: records, leaves and the parent's
// keys are independent).  rec_wait waits for EVERYTHING outstanding (vmcnt(0)); the "+v" operands tie the record's registers to it.
typedef float rec_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rec_load_agent_issue(const bvh2_node* n, rec_v4f& q0, rec_v4f& q1) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1" : "=&v"(q0), "=&v"(q1) : "v"(n) : "memory");
}
__device__ __forceinline__ void rec_wait(rec_v4f& q0, rec_v4f& q1) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1) :: "memory"); }
// box part only (bytes 8..31) of a node whose child links were written earlier: 8-byte + 16-byte write-through stores
__device__ __forceinline__ void node_box_store_agent(bvh2_node* n, const Box& b) {
    u64* q = reinterpret_cast<u64*>(n);
    st_agent(q + 1, pack2(b.lx, b.ly));
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f q1 = { b.lz, b.hx, b.hy, b.hz };
    asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1\n\ts_nop 1" :: "v"(n), "v"(q1) : "memory");
}
// whole node, plain (cached) stores: two 16-byte writes
__device__ __forceinline__ void node_store_plain(bvh2_node* n, u32 left, u32 right, const Box& b) {
    float4* q = reinterpret_cast<float4*>(n);
    q[0] = make_float4(__uint_as_float(left), __uint_as_float(right), b.lx, b.ly);
    q[1] = make_float4(b.lz, b.hx, b.hy, b.hz);
}
// plain loads / stores (data from an earlier kernel).  Aabb arrays (24-byte stride) and Bvh2Node::aabb (offset 8 of 32) are
// 8-byte aligned: three 8-byte accesses instead of six 4-byte ones.  PrimRef::aabb sits at offset 4 of a 28-byte record:
// box_load_u for those.
__device__ __forceinline__ Box box_load(const bvh_aabb* p) {
    const float2* f = reinterpret_cast<const float2*>(p);
    const float2 a = f[0], b = f[1], c = f[2];
    return { a.x, a.y, b.x, b.y, c.x, c.y };
}
// the gather form (one random 24-byte record per lane): a 16-byte + an 8-byte load instead of three 8-byte ones — the record is 8-byte aligned and a multi-dword
// global load only needs dword alignment, so the texture-address unit sees two lane requests per box instead of three
__device__ __forceinline__ Box box_gather(const bvh_aabb* p) {
    typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
    const f4a8 a = *reinterpret_cast<const f4a8*>(p);
    const float2 c = reinterpret_cast<const float2*>(p)[2];
    return { a.x, a.y, a.z, a.w, c.x, c.y };
}
// the bounds of a 64-byte Triangle record, with stage E's operations in stage E's order (stage_em.hip k_extents, including the clamp against the reset box:
// an all-NaN or all-+inf axis keeps +-FltMax): the same bits as the box array holds.  The clamps are integer-punned selects, not fminf / fmaxf: the emit kernels
// are compiled with -fno-honor-nans, under which a float minimum against a constant may be folded away.
__device__ __forceinline__ float clamp_lo_reset(float v) { return (v <= FMAX) ? v : FMAX; }        // fminf(FMAX, v): NaN and +inf -> FMAX
__device__ __forceinline__ float clamp_hi_reset(float v) { return (v >= -FMAX) ? v : -FMAX; }      // fmaxf(-FMAX, v): NaN and -inf -> -FMAX
__device__ __forceinline__ Box tri_box_gather(const float4* t) {
    const float4 a = t[0], b = t[1];
    const float c = reinterpret_cast<const float*>(t + 2)[0];
    return { clamp_lo_reset(fminf(fminf(a.x, a.w), b.z)), clamp_lo_reset(fminf(fminf(a.y, b.x), b.w)), clamp_lo_reset(fminf(fminf(a.z, b.y), c)),
             clamp_hi_reset(fmaxf(fmaxf(a.x, a.w), b.z)), clamp_hi_reset(fmaxf(fmaxf(a.y, b.x), b.w)), clamp_hi_reset(fmaxf(fmaxf(a.z, b.y), c)) };
}
__device__ __forceinline__ Box box_load_u(const bvh_aabb* p) {
    const float* f = reinterpret_cast<const float*>(p);
    return { f[0], f[1], f[2], f[3], f[4], f[5] };
}
__device__ __forceinline__ void box_store(bvh_aabb* p, const Box& b) {
    float2* f = reinterpret_cast<float2*>(p);
    f[0] = make_float2(b.lx, b.ly); f[1] = make_float2(b.lz, b.hx); f[2] = make_float2(b.hy, b.hz);
}

// ---- 16-lane DPP rows: lane l <- lane l + K of its row; lanes without a source read 0.  With two list slots per lane (slot 2l and 2l + 1) the
// neighbour of a slot at distance r is a row shift by r / 2 (or r / 2 + 1) lanes of the other (odd r) or the same (even r) register set, so a
// candidate union is six v_min / v_max with a DPP operand and no data movement at all.
template <int K> __device__ __forceinline__ float row_shl(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + K, 0xF, 0xF, true));
}
template <int K> __device__ __forceinline__ Box row_shl(const Box& b) {
    return { row_shl<K>(b.lx), row_shl<K>(b.ly), row_shl<K>(b.lz), row_shl<K>(b.hx), row_shl<K>(b.hy), row_shl<K>(b.hz) };
}
// whole-wave shift by one lane (DPP wave_shl:1): lane i <- lane i + 1, lane 63 reads 0
__device__ __forceinline__ float dpp_shl1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xF, 0xF, true));
}
__device__ __forceinline__ Box box_shl1(const Box& b) { return { dpp_shl1(b.lx), dpp_shl1(b.ly), dpp_shl1(b.lz), dpp_shl1(b.hx), dpp_shl1(b.hy), dpp_shl1(b.hz) }; }
__device__ __forceinline__ Box shfl_box(const Box& b, int src) {
    return { __shfl(b.lx, src), __shfl(b.ly, src), __shfl(b.lz, src), __shfl(b.hx, src), __shfl(b.hy, src), __shfl(b.hz, src) };
}
typedef float v2f_t __attribute__((ext_vector_type(2)));
// Aabb::area (src/Common.h:361-365) of two boxes at once (packed f32): 2 * (xy + xz + yz), same association, no contraction; x + x == 2 * x exactly
__device__ __forceinline__ v2f_t area_pair(v2f_t lx, v2f_t ly, v2f_t lz, v2f_t hx, v2f_t hy, v2f_t hz) {
    const v2f_t ex = hx - lx, ey = hy - ly, ez = hz - lz;
    const v2f_t h = ex * ey + ex * ez + ey * ez;
    return h + h;
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// The per-pass digit histograms of one key (k_morton / k_morton64 / k_hist): `passes` LDS atomics per lane.  A spatially coherent input (a real mesh's usual state; the
// benchmark's uniform mesh is the opposite) hands a wave 64 keys that share their high digits, and 64 LDS atomics on one address serialise: k_morton on the Morton-ordered
// 10 M mesh took 0.0912 ms against 0.0618 in random order, k_morton64 0.1205 against 0.0796, the stand-alone sort of sorted keys 0.286 against 0.241 (tools/ab_morton_order.py).
// Grouped counting: the lanes that hold the digit of the wave's first active lane are counted by ONE atomic of their lowest lane, the others add for themselves.  Measured
// forms (round 4, same tool; random / Morton order): behind divergent branches (mode 2) k_morton 0.0648 / 0.0640; branch-free (mode 3: every lane issues one ds_add, the
// group's other lanes add 0 to a word of their own) 0.0635 / 0.0630 — the default for the four digits of k_morton: +1.7 us on the benchmark's random order, -28 us on a
// coherent one; only when the wave's keys share their TOP digit (modes 1 / 4: one comparison, wave-uniform branch) 0.072 / 0.076 for k_morton (two code paths cost it more
// than they save) but right for the eight digits of k_morton64 (0.0806 / 0.098; always grouped: 0.094 / 0.094) and for k_hist (0.245 / 0.250).
#ifndef MORTON_GROUP
#define MORTON_GROUP 3       // k_morton (4 digits)
#endif
#ifndef MORTON64_GROUP
#define MORTON64_GROUP 4     // k_morton64 (8 digits)
#endif
#ifndef SORT_HIST_GROUP
#define SORT_HIST_GROUP 4    // k_hist (stand-alone sort)
#endif
// MODE: 0 = plain atomics, 1 = grouped when the top digit is shared, 2 = always grouped, 3 = always grouped and branch-free (every lane issues one ds_add: the group's other
// lanes add 0 to a word of their own in `pad`, 64 words), 4 = the branch-free form when the top digit is shared, plain atomics otherwise.
template <int MODE, typename DigitOf>
__device__ __forceinline__ void hist_add_passes(u32* hist, int passes, int stride, DigitOf digit_of, u32* pad = nullptr) {
    bool grouped = MODE == 2 || MODE == 3;
    if (MODE == 1 || MODE == 4) {
        const u32 top = digit_of(passes - 1);
        grouped = __ballot(top == (u32)__builtin_amdgcn_readfirstlane((int)top)) == __ballot(true);     // (uniform over the active lanes)
    }
    if (MODE == 3 || (MODE == 4 && grouped)) {
        const u32 lane = lane_id();
        for (int ps = 0; ps < passes; ++ps) {
            const u32 d = digit_of(ps);
            const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
            const u64 same = __ballot(d == d0);
            const bool mine = d == d0, lead = lane == (u32)__builtin_ctzll(same);
            u32* at = (mine && !lead) ? pad + lane : hist + ps * stride + d;
            atomicAdd(at, lead ? (u32)__popcll(same) : (mine ? 0u : 1u));
        }
    } else if (grouped) {
        const u64 lt = lanemask_lt();
        for (int ps = 0; ps < passes; ++ps) {
            const u32 d = digit_of(ps);
            const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)d);
            const u64 same = __ballot(d == d0);
            if (d != d0) atomicAdd(&hist[ps * stride + d], 1u);
            else if ((same & lt) == 0ull) atomicAdd(&hist[ps * stride + d0], (u32)__popcll(same));
        }
    } else {
        for (int ps = 0; ps < passes; ++ps) atomicAdd(&hist[ps * stride + digit_of(ps)], 1u);
    }
}

// 64-bit augmented key of sorted position i (Morton key in the high word, position in the low word)
__device__ __forceinline__ u64 aug_key(const u32* __restrict__ keys, u32 i) { return ((u64)keys[i] << 32) | i; }

// ---- key arithmetic of the hierarchy emitters, for 30-bit keys (u32, the reference) and 60-bit keys (u64).
// The emitters only ever ask (1) how many leading bits the augmented keys {Morton key, sorted position} of two positions
// share — plen, in [0, 32 + bits of K) — and (2) which of two adjacent pairs is closer.  For u32 keys the augmented key is one
// 64-bit word and "closer" is the reference's comparison of the xor values (src/SinglePassLbvhKernel.h:56-62,73); for u64 keys
// the augmented key has 96 bits and "closer" compares prefix lengths, which is the same order for the two boundary gaps of a
// range (they can never have equal prefix lengths: the keys inside the range share a strictly longer prefix than either).
__device__ __forceinline__ int clz_u64(u64 v) { return v ? __clzll((long long)v) : 64; }
__device__ __forceinline__ int clz_u32(u32 v) { return v ? __clz((int)v) : 32; }
__device__ __forceinline__ int plen(u32 ka, u32 i, u32 kb, u32 j) { return clz_u64((((u64)ka << 32) | i) ^ (((u64)kb << 32) | j)); }
__device__ __forceinline__ int plen(u64 ka, u32 i, u64 kb, u32 j) { return ka != kb ? clz_u64(ka ^ kb) : 64 + clz_u32(i ^ j); }
// do the augmented keys of positions i and j share (at least) their first c bits?  == (plen(...) >= c), without the count-leading-zeros:
// one 64-bit shift of the xor (u32 keys).  c == 0 (the root gap when adjacent keys differ in bit 31: possible for caller-supplied
// keys through bvh_emit_hploc, never for the 30-bit Morton codes) is an empty prefix, shared by everything: a shift by 64 is undefined
// (the hardware shifts by 0), so the shift is done in two steps that are each < 64
__device__ __forceinline__ bool shares_prefix(u32 ka, u32 i, u32 kb, u32 j, int c) {
    const u64 x = ((u64)(ka ^ kb) << 32) | (u64)(i ^ j);
    return ((x >> 1) >> (63 - c)) == 0ull;
}
__device__ __forceinline__ bool shares_prefix(u64 ka, u32 i, u64 kb, u32 j, int c) { return plen(ka, i, kb, j) >= c; }
template <typename K> struct KeyBits;
template <> struct KeyBits<u32> { static constexpr int value = 64; };    // bits of the augmented key = number of hierarchy levels
template <> struct KeyBits<u64> { static constexpr int value = 96; };
// is the pair (a, a+1) closer than the pair (b, b+1)?  (both pairs adjacent sorted positions)
__device__ __forceinline__ bool closer(const u32* __restrict__ k, u32 a, u32 b) { return (aug_key(k, a) ^ aug_key(k, a + 1)) < (aug_key(k, b) ^ aug_key(k, b + 1)); }
__device__ __forceinline__ bool closer(const u64* __restrict__ k, u32 a, u32 b) { return plen(k[a], a, k[a + 1], a + 1) > plen(k[b], b, k[b + 1], b + 1); }
// the same question with the four keys already loaded (ka = key[a], ka1 = key[a + 1], ...)
__device__ __forceinline__ bool closer_keys(u32 ka, u32 ka1, u32 a, u32 kb, u32 kb1, u32 b) {
    return ((((u64)ka << 32) | a) ^ (((u64)ka1 << 32) | (a + 1u))) < ((((u64)kb << 32) | b) ^ (((u64)kb1 << 32) | (b + 1u)));
}
__device__ __forceinline__ bool closer_keys(u64 ka, u64 ka1, u32 a, u64 kb, u64 kb1, u32 b) { return plen(ka, a, ka1, a + 1) > plen(kb, b, kb1, b + 1); }

} // namespace bvh
