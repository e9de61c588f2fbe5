// stage_em.hip — stage E (per-primitive AABB + scene extent) and stage M (30-bit extended Morton keys, with the
// radix-sort digit histograms fused in) for gfx950.
//
// E replaces CalculateSceneExtents (reference src/CommonBlocksKernel.h:92-114, helpers :27-78, float atomics
//   src/Common.h:291-307,400-408).  M replaces CalculateMortonCodes (src/CommonBlocksKernel.h:159-359,374-385).
// Both are HBM-streaming kernels: E moves 64 B in / 24 B out per primitive, M 24 B in / 4 B out.
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

// ------------------------------------------------------------------------------------------------------------------
// E: one thread per primitive, grid-stride.  The 64-byte Triangle record holds 9 floats; they are fetched as
// 2 x 16 B + 1 x 4 B so that no lane touches the 28 bytes of padding twice.  The block AABB is reduced with wave64
// shuffles + one LDS hop, then 6 integer-punned float atomics per block fold it into the scene extent.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {   // atomicMinFloat, src/Common.h:291-298
    if (__float_as_int(v) >= 0) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {   // atomicMaxFloat, src/Common.h:300-307
    if (__float_as_int(v) >= 0) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__device__ __forceinline__ Box wave_reduce_box(Box b) {
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) {
        b.lx = fminf(b.lx, __shfl_xor(b.lx, m)); b.ly = fminf(b.ly, __shfl_xor(b.ly, m)); b.lz = fminf(b.lz, __shfl_xor(b.lz, m));
        b.hx = fmaxf(b.hx, __shfl_xor(b.hx, m)); b.hy = fmaxf(b.hy, __shfl_xor(b.hy, m)); b.hz = fmaxf(b.hz, __shfl_xor(b.hz, m));
    }
    return b;
}

// block AABB -> scene extent: wave64 shuffles + one LDS hop + 6 integer-punned float atomics per block.  The stage-E kernels run
// EX_BLOCK = 1024 threads per workgroup on at most 512 workgroups: every workgroup ends with atomics on the same six words, one
// word takes ~90 atomics/us, and the workgroups of a streaming kernel all finish together — 2048 x 256-thread workgroups queued
// for 23 us at the end of a 10 M launch.
constexpr int EX_BLOCK = 1024;
__device__ __forceinline__ void block_reduce_scene(Box acc, float* __restrict__ scene) {
    acc = wave_reduce_box(acc);
    __shared__ float red[EX_BLOCK / WAVE][6];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    if (lane == 0) { red[wave][0] = acc.lx; red[wave][1] = acc.ly; red[wave][2] = acc.lz; red[wave][3] = acc.hx; red[wave][4] = acc.hy; red[wave][5] = acc.hz; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[0][threadIdx.x];
        for (int w = 1; w < nw; ++w) v = threadIdx.x < 3 ? fminf(v, red[w][threadIdx.x]) : fmaxf(v, red[w][threadIdx.x]);
        if (threadIdx.x < 3) atomic_min_f32(scene + threadIdx.x, v); else atomic_max_f32(scene + threadIdx.x, v);
    }
}

// The build path folds the per-build clearing (sort.hip k_prepare: digit histograms, look-back status rows, tile tickets / gate word, the emitters' queue
// heads) into its first kernel: every workgroup clears a slice before it touches a triangle — one launch (and its ~2 us boundary) less per build.  The
// scene extent cannot be reset here (this kernel's own atomics need it clean): the build path alternates between two extents and the Morton kernel of
// one build resets the extent of the next (api.hip build_impl).
__device__ __forceinline__ void prep_slice(const PrepArgs& p) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (u32 i = t; i < p.status_vecs; i += stride) p.status[i] = make_uint4(0u, 0u, 0u, 0u);
    for (u32 i = t; i < p.hist_words; i += stride) p.hist[i] = 0u;
    for (u32 i = t; i < p.extra_words; i += stride) p.extra[i] = 0u;
    if (p.counters && t < (u32)SORT_COUNTER_CLEAR) p.counters[t] = 0u;
    if (p.ploc_state) {                              // (block-uniform) PLOC++ build: what k_ploc_init clears
        for (u32 i = t; i < p.ploc_status_vecs; i += stride) p.ploc_status[i] = make_uint4(0u, 0u, 0u, 0u);
        if (t < p.ploc_tail_words) p.ploc_tail[t] = 0ull;
        if (t < (u32)PLOC_STATE_WORDS) p.ploc_state[t] = t == 0u ? p.ploc_count : 0u;
    }
}

__global__ __launch_bounds__(EX_BLOCK) void k_extents(const float4* __restrict__ tris, bvh_aabb* __restrict__ boxes,
                                                      float* __restrict__ scene, u32 n, PrepArgs prep) {
    prep_slice(prep);
    Box acc = box_empty();
    const u32 stride = gridDim.x * EX_BLOCK;
    for (u32 i = blockIdx.x * EX_BLOCK + threadIdx.x; i < n; i += stride) {
        const float4 a = tris[(size_t)i * 4 + 0];
        const float4 b = tris[(size_t)i * 4 + 1];
        const float  c = reinterpret_cast<const float*>(tris + (size_t)i * 4 + 2)[0];
        // v1 = (a.x,a.y,a.z)  v2 = (a.w,b.x,b.y)  v3 = (b.z,b.w,c)
        Box bx;
        // (Aabb() is the reset box and grow() is fminf / fmaxf, src/Common.h:327-345: a triangle whose three coordinates on an axis are all NaN — or all +inf — keeps
        // +-FltMax there; tests/test_gpu_round5.py ff_filled_triangle found the difference against the reference's kernel)
        bx.lx = fminf(FMAX, fminf(fminf(a.x, a.w), b.z)); bx.ly = fminf(FMAX, fminf(fminf(a.y, b.x), b.w)); bx.lz = fminf(FMAX, fminf(fminf(a.z, b.y), c));
        bx.hx = fmaxf(-FMAX, fmaxf(fmaxf(a.x, a.w), b.z)); bx.hy = fmaxf(-FMAX, fmaxf(fmaxf(a.y, b.x), b.w)); bx.hz = fmaxf(-FMAX, fmaxf(fmaxf(a.z, b.y), c));
        box_store(boxes + i, bx);
        acc = box_union(acc, bx);
    }
    block_reduce_scene(acc, scene);
}

// SURVEY.md §8(f) row 4 — device-side ingestion formats that do not pay for the reference's 64-byte padding.
// Packed: 9 floats per triangle (36-byte stride).  A block stages 256 triangles (9216 contiguous bytes) through LDS with
// 16-byte loads; each thread then reads its 9 floats at a stride of 9 words (odd: conflict-free).  R 36 + W 24 B / prim.
__global__ __launch_bounds__(EM_BLOCK) void k_extents_packed(const float* __restrict__ tris, bvh_aabb* __restrict__ boxes,
                                                             float* __restrict__ scene, u32 n, PrepArgs prep) {
    prep_slice(prep);
    __shared__ float s_t[EM_BLOCK * 9];
    Box acc = box_empty();
    const u32 tiles = (n + EM_BLOCK - 1) / EM_BLOCK;
    for (u32 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const u32 base = tile * EM_BLOCK;
        const u32 cnt = (n - base) < (u32)EM_BLOCK ? (n - base) : (u32)EM_BLOCK;
        const float* src = tris + (size_t)base * 9;                      // 16-byte aligned: EM_BLOCK * 36 is a multiple of 16
        const u32 words = cnt * 9u, vecs = words / 4u;
        for (u32 v = threadIdx.x; v < vecs; v += EM_BLOCK) reinterpret_cast<float4*>(s_t)[v] = reinterpret_cast<const float4*>(src)[v];
        for (u32 k = vecs * 4u + threadIdx.x; k < words; k += EM_BLOCK) s_t[k] = src[k];
        __syncthreads();
        if (threadIdx.x < cnt) {
            const float* t = s_t + threadIdx.x * 9;
            Box bx;
            bx.lx = fminf(FMAX, fminf(fminf(t[0], t[3]), t[6])); bx.ly = fminf(FMAX, fminf(fminf(t[1], t[4]), t[7])); bx.lz = fminf(FMAX, fminf(fminf(t[2], t[5]), t[8]));
            bx.hx = fmaxf(-FMAX, fmaxf(fmaxf(t[0], t[3]), t[6])); bx.hy = fmaxf(-FMAX, fmaxf(fmaxf(t[1], t[4]), t[7])); bx.hz = fmaxf(-FMAX, fmaxf(fmaxf(t[2], t[5]), t[8]));
            box_store(boxes + base + threadIdx.x, bx);
            acc = box_union(acc, bx);
        }
        __syncthreads();
    }
    block_reduce_scene(acc, scene);
}

// Indexed: float3 vertices + uint3 indices.  R 12 (indices) + 3 vertex gathers (12 B each, shared vertices hit in L2) + W 24 B / prim.
__global__ __launch_bounds__(EX_BLOCK) void k_extents_indexed(const float* __restrict__ verts, const u32* __restrict__ idx, u32 n_verts,
                                                              bvh_aabb* __restrict__ boxes, float* __restrict__ scene, u32 n, PrepArgs prep) {
    prep_slice(prep);
    Box acc = box_empty();
    const u32 stride = gridDim.x * EX_BLOCK;
    for (u32 i = blockIdx.x * EX_BLOCK + threadIdx.x; i < n; i += stride) {
        u32 i0 = idx[(size_t)i * 3 + 0], i1 = idx[(size_t)i * 3 + 1], i2 = idx[(size_t)i * 3 + 2];
        if (i0 >= n_verts) i0 = 0; if (i1 >= n_verts) i1 = 0; if (i2 >= n_verts) i2 = 0;   // never read out of bounds
        const float* a = verts + (size_t)i0 * 3; const float* b = verts + (size_t)i1 * 3; const float* c = verts + (size_t)i2 * 3;
        Box bx;
        bx.lx = fminf(FMAX, fminf(fminf(a[0], b[0]), c[0])); bx.ly = fminf(FMAX, fminf(fminf(a[1], b[1]), c[1])); bx.lz = fminf(FMAX, fminf(fminf(a[2], b[2]), c[2]));
        bx.hx = fmaxf(-FMAX, fmaxf(fmaxf(a[0], b[0]), c[0])); bx.hy = fmaxf(-FMAX, fmaxf(fmaxf(a[1], b[1]), c[1])); bx.hz = fmaxf(-FMAX, fmaxf(fmaxf(a[2], b[2]), c[2]));
        box_store(boxes + i, bx);
        acc = box_union(acc, bx);
    }
    block_reduce_scene(acc, scene);
}

__global__ void k_reset_scene(float* scene) {   // Aabb::reset on d_sceneExtents (src/PLOC++Bvh.cpp:23-25)
    if (threadIdx.x < 3) scene[threadIdx.x] = FMAX; else if (threadIdx.x < 6) scene[threadIdx.x] = -FMAX;
}

// ------------------------------------------------------------------------------------------------------------------
// M: the per-scene part of computeExtendedMortonCode (src/CommonBlocksKernel.h:162-275) depends only on the scene
// extent, so one lane per block evaluates it once ("plan") and broadcasts it through LDS; every thread then only
// quantises and interleaves.  Integer semantics follow the device code of the reference: float->int conversions
// saturate (v_cvt_i32_f32), mixed int/u32 min/max promote to double (value-exact), u32 arithmetic wraps.
// ------------------------------------------------------------------------------------------------------------------
struct MortonPlan { int axis[3]; int bits[3]; int pre[2]; int pre_sum; int swap; };

__device__ __forceinline__ int sat_f2i(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return -2147483647 - 1;
    return (int)f;
}
__device__ __forceinline__ u32 sat_f2u(float f) {
    if (f != f) return 0u;
    if (f <= 0.0f) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (u32)f;
}
__device__ __forceinline__ int lg_ratio(float num, float den) { return sat_f2i(log2f(num / den)); }

// NB: total bit budget of the code — 30 in the reference (:161); 60 for the u64 keys of SURVEY.md §8(f) row 3 (same arithmetic)
__device__ void make_plan(const float* __restrict__ scene, MortonPlan& m, float* lo, float* ext, const u32 NB = 30u) {
    lo[0] = scene[0]; lo[1] = scene[1]; lo[2] = scene[2];
    const float ex = scene[3] - scene[0], ey = scene[4] - scene[1], ez = scene[5] - scene[2];
    ext[0] = ex; ext[1] = ey; ext[2] = ez;
    int px, py, pz;
    // axis order by extent; the strict '<' chain of src/CommonBlocksKernel.h:167-250 decides ties
    if (ex < ey) {
        if (ex < ez) {
            if (ey < ez) { m.axis[0] = 2; m.axis[1] = 1; m.axis[2] = 0; px = lg_ratio(ez, ey); py = lg_ratio(ey, ex); pz = lg_ratio(ez, ex); }
            else         { m.axis[0] = 1; m.axis[1] = 2; m.axis[2] = 0; px = lg_ratio(ey, ez); py = lg_ratio(ez, ex); pz = lg_ratio(ey, ex); }
        } else           { m.axis[0] = 1; m.axis[1] = 0; m.axis[2] = 2; px = lg_ratio(ey, ex); py = lg_ratio(ex, ez); pz = lg_ratio(ey, ez); }
    } else {
        if (ey < ez) {
            if (ex < ez) { m.axis[0] = 2; m.axis[1] = 0; m.axis[2] = 1; px = lg_ratio(ez, ex); py = lg_ratio(ex, ey); pz = lg_ratio(ez, ey); }
            else         { m.axis[0] = 0; m.axis[1] = 2; m.axis[2] = 1; px = lg_ratio(ex, ez); py = lg_ratio(ez, ey); pz = lg_ratio(ex, ey); }
        } else           { m.axis[0] = 0; m.axis[1] = 1; m.axis[2] = 2; px = lg_ratio(ex, ey); py = lg_ratio(ey, ez); pz = lg_ratio(ex, ez); }
    }
    int swap = (int)((u32)pz - ((u32)px + (u32)py));                                   // :252
    px = (int)fmin((double)px, (double)NB);                                            // :254
    py = (int)(fmin((double)(int)((u32)py * 2u), (double)(NB - (u32)px)) / 2.0);       // :255
    int sum = (int)((u32)px + (u32)py * 2u);                                           // :257
    if (sum != (int)NB) sum = (int)((u32)sum + (u32)swap); else swap = 0;              // :259-262
    const int bz = (ext[m.axis[2]] != 0.0f) ? (int)fmax(0.0, (double)((NB - (u32)sum) / 3u)) : 0;   // :264
    int bx, by;
    if (swap > 0) { bx = (int)fmax(0.0, (double)((NB - (u32)bz - (u32)sum) / 2u + (u32)py + (u32)px + 1u)); by = (int)(NB - (u32)bx - (u32)bz); }   // :266-270
    else          { by = (int)fmax(0.0, (double)((NB - (u32)bz - (u32)sum) / 2u + (u32)py));                 bx = (int)(NB - (u32)by - (u32)bz); }   // :271-275
    m.bits[0] = bx; m.bits[1] = by; m.bits[2] = bz; m.pre[0] = px; m.pre[1] = py; m.pre_sum = sum; m.swap = swap;
}

__device__ __forceinline__ u32 spread2(u32 v) {   // morton2D, :139-147
    v &= 0x0000ffffu; v = (v ^ (v << 8)) & 0x00ff00ffu; v = (v ^ (v << 4)) & 0x0f0f0f0fu;
    v = (v ^ (v << 2)) & 0x33333333u; v = (v ^ (v << 1)) & 0x55555555u; return v;
}
__device__ __forceinline__ u32 spread3(u32 x) {   // morton3D, :149-156
    x = (x * 0x00010001u) & 0xFF0000FFu; x = (x * 0x00000101u) & 0x0F00F00Fu;
    x = (x * 0x00000011u) & 0xC30C30C3u; x = (x * 0x00000005u) & 0x49249249u; return x;
}
__device__ __forceinline__ u32 shl(u32 v, u32 s) { return s >= 32u ? 0u : v << s; }
__device__ __forceinline__ u32 shr(u32 v, u32 s) { return s >= 32u ? 0u : v >> s; }

__device__ __forceinline__ u32 encode(const MortonPlan& m, float p0, float p1, float p2) {   // :277-358; p_k = position on axis[k]
    int bx = m.bits[0], by = m.bits[1];
    const int bz = m.bits[2], px = m.pre[0], py = m.pre[1];
    u32 q0 = min(sat_f2u(fmaxf(p0 * (float)shl(1u, (u32)bx), 0.0f)), shl(1u, (u32)bx) - 1u);
    u32 q1 = min(sat_f2u(fmaxf(p1 * (float)shl(1u, (u32)by), 0.0f)), shl(1u, (u32)by) - 1u);
    u32 q2 = min(sat_f2u(fmaxf(p2 * (float)shl(1u, (u32)bz), 0.0f)), shl(1u, (u32)bz) - 1u);
    u32 code = 0, d0 = 0, d1 = 0;
    if (m.pre_sum > 0) {
        bx -= px;
        code = shr(q0 & shl(shl(1u, (u32)px) - 1u, (u32)bx), (u32)bx);
        code = shl(code, (u32)(py * 2));
        bx -= py; by -= py;
        const u32 t0 = spread2(shr(q0 & shl(shl(1u, (u32)py) - 1u, (u32)bx), (u32)bx));
        const u32 t1 = spread2(shr(q1 & shl(shl(1u, (u32)py) - 1u, (u32)by), (u32)by));
        code |= t0 * 2 + t1;
        if (m.swap > 0) { code <<= 1; bx -= 1; code |= shr(q0 & shl(1u, (u32)bx), (u32)bx); }
        code = shl(code, (u32)(bx + by + bz));
        q0 &= shl(1u, (u32)bx) - 1u;
        q1 &= shl(1u, (u32)by) - 1u;
        if (m.swap > 0) { d0 = (u32)(by - bx); q0 = shl(q0, d0); d1 = (u32)(by - bz); q2 = shl(q2, d1); }
        else            { d0 = (u32)(bx - by); q1 = shl(q1, d0); d1 = (u32)(bx - bz); q2 = shl(q2, d1); }
    }
    if (bz == 0) code |= spread2(q0) * 2 + spread2(q1);
    else {
        const u32 X = spread3(q0), Y = spread3(q1), Z = spread3(q2);
        code |= shr((m.swap > 0) ? (Y * 4 + X * 2 + Z) : (X * 4 + Y * 2 + Z), d0 + d1);
    }
    return code;
}

// ---- 64-bit flavour of encode(): the same steps with every intermediate 64 bits wide, for bit budgets up to 60 (<= 20 bits per
// axis in the 3-D part, <= 30 in the 2-D parts).  With a 30-bit budget it reproduces encode() bit for bit (tests pin that).
__device__ __forceinline__ u64 shl64(u64 v, u32 s) { return s >= 64u ? 0ull : v << s; }
__device__ __forceinline__ u64 shr64(u64 v, u32 s) { return s >= 64u ? 0ull : v >> s; }
__device__ __forceinline__ u64 spread2_64(u64 v) {   // bit i -> bit 2i, 32-bit input
    v &= 0x00000000ffffffffull; v = (v ^ (v << 16)) & 0x0000ffff0000ffffull; v = (v ^ (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v ^ (v << 4)) & 0x0f0f0f0f0f0f0f0full; v = (v ^ (v << 2)) & 0x3333333333333333ull; v = (v ^ (v << 1)) & 0x5555555555555555ull; return v;
}
__device__ __forceinline__ u64 spread3_64(u64 x) {   // bit i -> bit 3i, 21-bit input
    x &= 0x1fffffull; x = (x | (x << 32)) & 0x1f00000000ffffull; x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full; x = (x | (x << 4)) & 0x10c30c30c30c30c3ull; x = (x | (x << 2)) & 0x1249249249249249ull; return x;
}
__device__ __forceinline__ u64 sat_f2u64(float f) {
    if (f != f) return 0ull;
    if (f <= 0.0f) return 0ull;
    if (f >= 18446744073709551616.0f) return ~0ull;
    return (u64)f;
}
__device__ __forceinline__ u64 quantise64(float p, int bits) {
    const u64 top = shl64(1ull, (u32)bits);
    const u64 q = sat_f2u64(fmaxf(p * (float)top, 0.0f));
    return q < top - 1ull ? q : top - 1ull;
}
__device__ __forceinline__ u64 encode64(const MortonPlan& m, float p0, float p1, float p2) {
    int bx = m.bits[0], by = m.bits[1];
    const int bz = m.bits[2], px = m.pre[0], py = m.pre[1];
    u64 q0 = quantise64(p0, bx), q1 = quantise64(p1, by), q2 = quantise64(p2, bz);
    u64 code = 0; u32 d0 = 0, d1 = 0;
    if (m.pre_sum > 0) {
        bx -= px;
        code = shr64(q0 & shl64(shl64(1ull, (u32)px) - 1ull, (u32)bx), (u32)bx);
        code = shl64(code, (u32)(py * 2));
        bx -= py; by -= py;
        const u64 t0 = spread2_64(shr64(q0 & shl64(shl64(1ull, (u32)py) - 1ull, (u32)bx), (u32)bx));
        const u64 t1 = spread2_64(shr64(q1 & shl64(shl64(1ull, (u32)py) - 1ull, (u32)by), (u32)by));
        code |= t0 * 2 + t1;
        if (m.swap > 0) { code <<= 1; bx -= 1; code |= shr64(q0 & shl64(1ull, (u32)bx), (u32)bx); }
        code = shl64(code, (u32)(bx + by + bz));
        q0 &= shl64(1ull, (u32)bx) - 1ull;
        q1 &= shl64(1ull, (u32)by) - 1ull;
        if (m.swap > 0) { d0 = (u32)(by - bx); q0 = shl64(q0, d0); d1 = (u32)(by - bz); q2 = shl64(q2, d1); }
        else            { d0 = (u32)(bx - by); q1 = shl64(q1, d0); d1 = (u32)(bx - bz); q2 = shl64(q2, d1); }
    }
    if (bz == 0) code |= spread2_64(q0) * 2 + spread2_64(q1);
    else {
        const u64 X = spread3_64(q0), Y = spread3_64(q1), Z = spread3_64(q2);
        code |= shr64((m.swap > 0) ? (Y * 4 + X * 2 + Z) : (X * 4 + Y * 2 + Z), d0 + d1);
    }
    return code;
}

// u64 keys with a `total_bits` budget; the 8-bit digit histograms of the `passes` sort passes that follow are fused in as in k_morton
__global__ __launch_bounds__(EM_BLOCK) void k_morton64(const bvh_aabb* __restrict__ boxes, const float* __restrict__ scene,
                                                       u64* __restrict__ keys, u32 n, u32 total_bits, u32* __restrict__ hist, int passes, float* reset_next) {
    if (reset_next && blockIdx.x == 0 && threadIdx.x < 6) reset_next[threadIdx.x] = threadIdx.x < 3 ? FMAX : -FMAX;     // Aabb::reset of the NEXT build's extent
    __shared__ MortonPlan s_plan; __shared__ float s_lo[3], s_ext[3];
    __shared__ u32 s_hist[8 * 256];
    __shared__ u32 s_pad[MORTON64_GROUP >= 3 ? 64 : 1];
    if (threadIdx.x == 0) make_plan(scene, s_plan, s_lo, s_ext, total_bits);
    for (int i = threadIdx.x; i < passes * 256; i += EM_BLOCK) s_hist[i] = 0;
    __syncthreads();
    const MortonPlan m = s_plan;
    const float lo[3] = { s_lo[0], s_lo[1], s_lo[2] }, ext[3] = { s_ext[0], s_ext[1], s_ext[2] };
    const u32 ntile = (n + EM_BLOCK - 1) / EM_BLOCK;                      // tiles in descending order, as in k_morton
    for (u32 tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const u32 i = (ntile - 1u - tile) * EM_BLOCK + threadIdx.x;
        if (i >= n) continue;
        const Box b = box_load(boxes + i);
        const float p[3] = { ((b.hx + b.lx) * 0.5f - lo[0]) / ext[0], ((b.hy + b.ly) * 0.5f - lo[1]) / ext[1], ((b.hz + b.lz) * 0.5f - lo[2]) / ext[2] };
        const u64 code = encode64(m, p[m.axis[0]], p[m.axis[1]], p[m.axis[2]]);
        keys[i] = code;
        hist_add_passes<MORTON64_GROUP>(s_hist, passes, 256, [&](int ps) { return (u32)(code >> (ps * 8)) & 255u; }, s_pad);
    }
    __syncthreads();
    u32* copy = hist + (blockIdx.x % SORT_HIST_COPIES) * SORT_HIST_STRIDE;
    for (int i = threadIdx.x; i < passes * 256; i += EM_BLOCK) { const u32 c = s_hist[i]; if (c) atomicAdd(&copy[i], c); }
}

// HIST_BITS > 0: also accumulate the per-pass digit histograms of the LSD radix sort that follows (digits of HIST_BITS
// bits starting at bit 0, `passes` of them) — LDS histogram per block, flushed with one global atomic per non-empty bin.
template <int HIST_BITS>
__global__ __launch_bounds__(EM_BLOCK) void k_morton(const bvh_aabb* __restrict__ boxes, const float* __restrict__ scene,
                                                     u32* __restrict__ keys, u32* __restrict__ vals, u32 n,
                                                     u32* __restrict__ hist, int passes, float* reset_next) {
    if (reset_next && blockIdx.x == 0 && threadIdx.x < 6) reset_next[threadIdx.x] = threadIdx.x < 3 ? FMAX : -FMAX;     // Aabb::reset of the NEXT build's extent
    __shared__ MortonPlan s_plan; __shared__ float s_lo[3], s_ext[3];
    constexpr int RADIX = HIST_BITS > 0 ? (1 << HIST_BITS) : 1;
    __shared__ u32 s_hist[HIST_BITS > 0 ? 4 * RADIX : 1];
    __shared__ u32 s_pad[MORTON_GROUP >= 3 ? 64 : 1];
    if (threadIdx.x == 0) make_plan(scene, s_plan, s_lo, s_ext);
    if (HIST_BITS > 0) for (int i = threadIdx.x; i < passes * RADIX; i += EM_BLOCK) s_hist[i] = 0;
    __syncthreads();
    const MortonPlan m = s_plan;
    const float lo[3] = { s_lo[0], s_lo[1], s_lo[2] }, ext[3] = { s_ext[0], s_ext[1], s_ext[2] };
    // tiles in descending order: stage E wrote the boxes in ascending order just before, so the last ones are the ones still in the caches
    const u32 ntile = (n + EM_BLOCK - 1) / EM_BLOCK;
    for (u32 tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const u32 i = (ntile - 1u - tile) * EM_BLOCK + threadIdx.x;
        if (i >= n) continue;
        const Box b = box_load(boxes + i);
        // centre = (max + min) * 0.5f (src/Common.h:347); p = (centre - scene.min) / extent with IEEE divides (:381)
        const float p[3] = { ((b.hx + b.lx) * 0.5f - lo[0]) / ext[0], ((b.hy + b.ly) * 0.5f - lo[1]) / ext[1], ((b.hz + b.lz) * 0.5f - lo[2]) / ext[2] };
        const u32 code = encode(m, p[m.axis[0]], p[m.axis[1]], p[m.axis[2]]);
        keys[i] = code;
        if (vals) vals[i] = i;                                   // :384
        if (HIST_BITS > 0) {
            hist_add_passes<MORTON_GROUP>(s_hist, passes, RADIX, [&](int ps) { return (code >> (ps * HIST_BITS)) & (u32)(RADIX - 1); }, s_pad);
        }
    }
    if (HIST_BITS > 0) {
        __syncthreads();
        u32* copy = hist + (blockIdx.x % SORT_HIST_COPIES) * SORT_HIST_STRIDE;      // (kernels.hpp: why there are several copies)
        for (int i = threadIdx.x; i < passes * RADIX; i += EM_BLOCK) {
            const u32 c = s_hist[i];
            if (c) {
                atomicAdd(&copy[i], c);
                // a code with bit 30 or 31 set (top digit >= 64): the build's sort must take the full-width top pass (SORT_WIDE_FLAG_WORD, kernels.hpp)
                if (HIST_BITS == 8 && passes == 4 && i >= 3 * RADIX + 64) hist[SORT_WIDE_FLAG_WORD] = 1u;
            }
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------
#ifndef EM_PPT
#define EM_PPT 4         // primitives per thread of the grid-stride kernels below their workgroup caps (A/B: 2 / 1 measured at 262 144, LEADS.md)
#endif
#ifndef EX_PPT
#define EX_PPT 4
#endif
static inline int em_grid(u32 n) {
    // grid-stride kernels: ~4 primitives per thread up to 2048 workgroups (256 CUs x 8).  Every workgroup ends with global atomics on a
    // handful of addresses (6 scene-extent words; <= 1024 histogram bins), and one address takes ~90 atomics/us: at 262 k primitives
    // 1024 one-tile workgroups spent 11 us in that queue (k_extents 31 us, k_morton 23 us), 256 four-tile workgroups do not.
    const u32 blocks = (n + EM_PPT * EM_BLOCK - 1) / (EM_PPT * EM_BLOCK);
    return (int)(blocks < 2048u ? (blocks ? blocks : 1u) : 2048u);
}

static inline int ex_grid(u32 n) {            // ~4 primitives per thread, at most 2 workgroups of 1024 threads per CU
    const u32 blocks = (n + EX_PPT * EX_BLOCK - 1) / (EX_PPT * EX_BLOCK);
    return (int)(blocks < 512u ? (blocks ? blocks : 1u) : 512u);
}

void launch_extents(hipStream_t s, const void* d_tris, u32 n, void* d_boxes, void* d_scene, bool reset_scene, const PrepArgs* prep) {
    const PrepArgs pa = prep ? *prep : PrepArgs{};
    if (reset_scene) hipLaunchKernelGGL(k_reset_scene, dim3(1), dim3(64), 0, s, (float*)d_scene);
    // (a variant in which four lanes read one 64-byte record with 16-byte loads and three of them store the box — every access fully
    // coalesced — runs in the same time: 0.180 vs 0.179 ms at 10 M, the kernel moves 880 MB at 4.9 TB/s either way)
    KernelScope ks(s, "k_extents");
    hipLaunchKernelGGL(k_extents, dim3(ex_grid(n)), dim3(EX_BLOCK), 0, s, (const float4*)d_tris, (bvh_aabb*)d_boxes, (float*)d_scene, n, pa);
}

void launch_extents_packed(hipStream_t s, const void* d_tris36, u32 n, void* d_boxes, void* d_scene, bool reset_scene, const PrepArgs* prep) {
    const PrepArgs pa = prep ? *prep : PrepArgs{};
    if (reset_scene) hipLaunchKernelGGL(k_reset_scene, dim3(1), dim3(64), 0, s, (float*)d_scene);
    KernelScope ks(s, "k_extents_packed");
    hipLaunchKernelGGL(k_extents_packed, dim3(em_grid(n)), dim3(EM_BLOCK), 0, s, (const float*)d_tris36, (bvh_aabb*)d_boxes, (float*)d_scene, n, pa);
}
void launch_extents_indexed(hipStream_t s, const void* d_vertices, const void* d_indices, u32 n_vertices, u32 n, void* d_boxes, void* d_scene, bool reset_scene, const PrepArgs* prep) {
    const PrepArgs pa = prep ? *prep : PrepArgs{};
    if (reset_scene) hipLaunchKernelGGL(k_reset_scene, dim3(1), dim3(64), 0, s, (float*)d_scene);
    KernelScope ks(s, "k_extents_indexed");
    hipLaunchKernelGGL(k_extents_indexed, dim3(ex_grid(n)), dim3(EX_BLOCK), 0, s, (const float*)d_vertices, (const u32*)d_indices, n_vertices, (bvh_aabb*)d_boxes, (float*)d_scene, n, pa);
}

// the per-scene plan of the Morton kernels, as the device evaluates it (bvh_stage_morton_plan): {axis[3], bits[3], pre[2], pre_sum, swap}
__global__ void k_morton_plan(const float* __restrict__ scene, int* __restrict__ out, u32 total_bits) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    MortonPlan m; float lo[3], ext[3];
    make_plan(scene, m, lo, ext, total_bits);
    for (int i = 0; i < 3; ++i) { out[i] = m.axis[i]; out[3 + i] = m.bits[i]; }
    out[6] = m.pre[0]; out[7] = m.pre[1]; out[8] = m.pre_sum; out[9] = m.swap;
}
void launch_morton_plan(hipStream_t s, const void* d_scene, int* d_out, int total_bits) {
    hipLaunchKernelGGL(k_morton_plan, dim3(1), dim3(64), 0, s, (const float*)d_scene, d_out, (u32)total_bits);
}

void launch_morton(hipStream_t s, const void* d_boxes, u32 n, const void* d_scene, u32* d_keys, u32* d_vals,
                   u32* d_hist, int hist_bits, int passes, float* d_reset_next) {
    const dim3 g(em_grid(n)), b(EM_BLOCK);
    KernelScope ks(s, "k_morton");
    if (d_hist && hist_bits == 8)       hipLaunchKernelGGL(k_morton<8>,  g, b, 0, s, (const bvh_aabb*)d_boxes, (const float*)d_scene, d_keys, d_vals, n, d_hist, passes, d_reset_next);
    else                                hipLaunchKernelGGL(k_morton<0>,  g, b, 0, s, (const bvh_aabb*)d_boxes, (const float*)d_scene, d_keys, d_vals, n, (u32*)nullptr, 0, d_reset_next);
}

void launch_morton64(hipStream_t s, const void* d_boxes, u32 n, const void* d_scene, uint64_t* d_keys, int total_bits, u32* d_hist, int passes, float* d_reset_next) {
    KernelScope ks(s, "k_morton64");
    hipLaunchKernelGGL(k_morton64, dim3(em_grid(n)), dim3(EM_BLOCK), 0, s, (const bvh_aabb*)d_boxes, (const float*)d_scene, d_keys, n, (u32)total_bits,
                       d_hist, d_hist ? passes : 0, d_reset_next);
}

// (kernels.hpp: touching one kernel of this translation unit makes the runtime load its code object — bvh_ctx_create does that for the build path's modules, so
// that a context's FIRST build does not pay for it: 0.3-0.7 ms per module on the MI355X, tools/cold_probe.py)
void warm_stage_em() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_reset_scene)); }

} // namespace bvh
