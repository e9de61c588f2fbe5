// bvh_oracle.cpp — CPU restatement of the reference's BVH build hot path.
//
// *** TEST INFRASTRUCTURE ONLY ***  Nothing under oracle/ is part of the product.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
// checker.  The product (hip-bvh-construction_amd/csrc) never links or calls it.
//
// Every function cites the reference file:line (relative to the reference repo root) it follows.
// This is a restatement (own structure, serial, host-only), not a copy: the reference code is
// GPU kernels; here each kernel becomes a plain loop with the same arithmetic in the same order.
//
// Pinning status (see DESIGN.md "Oracle"): the reference ships no tests or golden vectors for this
// path.  The oracle is pinned against (1) the reference's own Utility.cpp (validators + SAH cost)
// built unmodified into oracle/_ref/libref_utility.so, and (2) the reference's own kernels compiled
// unmodified by hipcc into oracle/_ref/*.co and executed on the MI355X by oracle/ref_driver.cpp —
// tests/test_reference_kernels.py compares them with this file on the GPU box — and (3) the reference's PLOC++ kernels (SetupClusters, Ploc,
// SinglePassPloc) and both CollapseToWide4Bvh kernels executed on the CPU under the fiber SIMT emulator of tools/oracle/ref_emulator.cpp
// (oracle/_ref/libref_ploc_emu.so, libref_lbvh_emu.so; outputs committed in tests/golden/reference_outputs.json): orc_ploc's node arrays
// equal the emulated reference's byte for byte, orc_collapse4's wide trees its topology and cost — and (4, round 5) the reference's own wave64 build of
// CalculateSceneExtents, Ploc, SinglePassPloc and both CollapseToWide4Bvh kernels executed on the MI355X (oracle/_ref/*.w64.co, `_ploc_hw` goldens,
// tests/test_reference_w64.py, tests/test_oracle_golden.py).  The radix sort
// (Orochi, un-vendored submodule, version unknown) has no reference-side pin: "parity unpinned" for
// the sort boundary; the contract adopted is a stable ascending sort of the 32-bit key.
//
// Build: make -C oracle   (g++ -O2 -ffp-contract=off; no dependencies)

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <queue>
#include <vector>

namespace {

using u32 = uint32_t;
using u64 = uint64_t;
constexpr u32 INV = 0xFFFFFFFFu;           // src/Common.h:90-92
constexpr float FLT_MAX_ = 3.402823466e+38f;  // src/Common.h:86

struct F3 { float x, y, z; };
struct Box { F3 lo, hi; };                                   // src/Common.h:310-416 (24 B)
struct alignas(64) Tri { F3 a, b, c; };                      // src/Common.h:429-434 (64 B)
struct alignas(32) Node2 { u32 l, r; Box b; };               // src/Common.h:436-441 (32 B)
struct Leaf { u32 prim; Box b; };                            // src/Common.h:574-578 (28 B)
struct alignas(32) SahNode { Box b; u32 first; u32 count; }; // src/Common.h:443-453 (32 B)
struct alignas(128) Node4 { Box b[4]; u32 child[4]; u32 parent; u32 count; };  // src/Common.h:560-566
struct PrimNode { u32 prim; u32 parent; };                   // src/Common.h:568-572

static_assert(sizeof(Box) == 24 && sizeof(Tri) == 64 && sizeof(Node2) == 32 && sizeof(Leaf) == 28, "layout");
static_assert(sizeof(SahNode) == 32 && sizeof(Node4) == 128 && sizeof(PrimNode) == 8, "layout");

inline Box box_empty() { return { {FLT_MAX_, FLT_MAX_, FLT_MAX_}, {-FLT_MAX_, -FLT_MAX_, -FLT_MAX_} }; }  // Common.h:327-331
inline void grow(Box& b, const F3& p) {                                                            // Common.h:340-345
    b.lo = { fminf(b.lo.x, p.x), fminf(b.lo.y, p.y), fminf(b.lo.z, p.z) };
    b.hi = { fmaxf(b.hi.x, p.x), fmaxf(b.hi.y, p.y), fmaxf(b.hi.z, p.z) };
}
inline Box unite(const Box& a, const Box& b) {                                                     // Common.h:333-338,456-459
    return { { fminf(a.lo.x, b.lo.x), fminf(a.lo.y, b.lo.y), fminf(a.lo.z, b.lo.z) },
             { fmaxf(a.hi.x, b.hi.x), fmaxf(a.hi.y, b.hi.y), fmaxf(a.hi.z, b.hi.z) } };
}
inline float area(const Box& b) {                                                                  // Common.h:361-365
    const float ex = b.hi.x - b.lo.x, ey = b.hi.y - b.lo.y, ez = b.hi.z - b.lo.z;
    return 2 * (ex * ey + ex * ez + ey * ez);   // this association, no FMA (built with -ffp-contract=off)
}
inline Box tri_box(const Tri& t) { Box b = box_empty(); grow(b, t.a); grow(b, t.b); grow(b, t.c); return b; }
inline u32 fbits(float f) { u32 u; std::memcpy(&u, &f, 4); return u; }

// Float -> int conversions as the AMD VALU performs them (v_cvt_i32_f32 / v_cvt_u32_f32 saturate, NaN -> 0).
// The reference relies on them implicitly for degenerate extents (Appendix B of SURVEY.md).
inline int sat_f2i(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return -2147483647 - 1;
    return (int)f;
}
inline u32 sat_f2u(float f) {
    if (f != f) return 0u;
    if (f <= 0.0f) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (u32)f;
}
inline int wrap_add(int a, int b) { return (int)((u32)a + (u32)b); }
inline int wrap_sub(int a, int b) { return (int)((u32)a - (u32)b); }
inline int wrap_mul2(int a) { return (int)((u32)a * 2u); }

// ------------------------------------------------------------------------------------------------
// Stage M — extended 30-bit Morton code.  src/CommonBlocksKernel.h:159-359.
// Everything that depends only on the scene extent (:162-275) is hoisted into MortonPlan.
// ------------------------------------------------------------------------------------------------
struct MortonPlan {
    int axis[3];     // startAxis.{x,y,z}        (:167-250)
    int bits[3];     // numBits.{x,y,z}          (:264-275)
    int pre[2];      // numPrebits.{x,y} after clamping (:254-255)
    int pre_sum;     // numPrebitsSum            (:257-262)
    int swap;        // swap                     (:252,259-262)
};

// mixed int/u32 min/max in HIP device code promote to double; value-exact.
inline double dmin(double a, double b) { return a < b ? a : b; }
inline double dmax(double a, double b) { return a > b ? a : b; }

MortonPlan morton_plan(const F3& e, const u32 NB = 30) {   // NB: bit budget — 30 in the reference (:161); 60 = the u64 extension
    MortonPlan m;
    const float ext[3] = { e.x, e.y, e.z };
    int px, py, pz;
    // (int)log2f(ratio), :175-248.  The reference evaluates it on the DEVICE: OCML's log2f (v_log_f32: exponent + log2(mantissa), the fraction always < 1) never
    // rounds a ratio just below 2^k up to k, whereas a correctly rounded host log2f does from k = 2 on (log2(2^k (1 - 2^-24)) = k - 8.6e-8 is nearer to k than to the
    // float below k).  Measured on the MI355X (tests/test_gpu_round4.py::test_morton_plan_at_log2_boundaries, round 4): with the host's libm here, 228 of 720 scenes whose
    // extent ratios sit 1-2 ulp below a power of two got a different bit plan than the reference's own CalculateMortonCodes kernel; the truncated value the device
    // produces is the exact floor, i.e. the ratio's binary exponent.  So: finite ratios >= 1 (the if-chain below always divides the larger extent by the smaller)
    // take ilogbf; everything else (a zero extent: inf or NaN) goes through log2f and the saturating conversion as before.
    auto lg = [](float num, float den) {
        const float r = num / den;
        if (r >= 1.0f && r <= FLT_MAX) return (int)ilogbf(r);
        return sat_f2i(log2f(r));
    };
    // literal if-chain of :167-250 (strict '<' comparisons decide tie order)
    if (e.x < e.y) {
        if (e.x < e.z) {
            if (e.y < e.z) { m.axis[0] = 2; m.axis[1] = 1; m.axis[2] = 0; px = lg(e.z, e.y); py = lg(e.y, e.x); pz = lg(e.z, e.x); }
            else           { m.axis[0] = 1; m.axis[1] = 2; m.axis[2] = 0; px = lg(e.y, e.z); py = lg(e.z, e.x); pz = lg(e.y, e.x); }
        } else             { m.axis[0] = 1; m.axis[1] = 0; m.axis[2] = 2; px = lg(e.y, e.x); py = lg(e.x, e.z); pz = lg(e.y, e.z); }
    } else {
        if (e.y < e.z) {
            if (e.x < e.z) { m.axis[0] = 2; m.axis[1] = 0; m.axis[2] = 1; px = lg(e.z, e.x); py = lg(e.x, e.y); pz = lg(e.z, e.y); }
            else           { m.axis[0] = 0; m.axis[1] = 2; m.axis[2] = 1; px = lg(e.x, e.z); py = lg(e.z, e.y); pz = lg(e.x, e.y); }
        } else             { m.axis[0] = 0; m.axis[1] = 1; m.axis[2] = 2; px = lg(e.x, e.y); py = lg(e.y, e.z); pz = lg(e.x, e.z); }
    }
    int swap = wrap_sub(pz, wrap_add(px, py));                      // :252
    px = (int)dmin((double)px, (double)NB);                         // :254
    py = (int)(dmin((double)wrap_mul2(py), (double)(u32)(NB - (u32)px)) / 2);   // :255 (double divide, then trunc)
    int sum = wrap_add(px, wrap_mul2(py));                          // :257
    if (sum != (int)NB) sum = wrap_add(sum, swap); else swap = 0;   // :259-262
    int bz = (ext[m.axis[2]] != 0) ? (int)dmax(0.0, (double)((u32)(NB - (u32)sum) / 3u)) : 0;   // :264 (u32 arithmetic)
    int bx, by;
    if (swap > 0) {                                                 // :266-270
        bx = (int)dmax(0.0, (double)(u32)((NB - (u32)bz - (u32)sum) / 2u + (u32)py + (u32)px + 1u));
        by = (int)(NB - (u32)bx - (u32)bz);
    } else {                                                        // :271-275
        by = (int)dmax(0.0, (double)(u32)((NB - (u32)bz - (u32)sum) / 2u + (u32)py));
        bx = (int)(NB - (u32)by - (u32)bz);
    }
    m.bits[0] = bx; m.bits[1] = by; m.bits[2] = bz;
    m.pre[0] = px; m.pre[1] = py; m.pre_sum = sum; m.swap = swap;
    return m;
}

inline u32 spread2(u32 v) {   // morton2D :139-147
    v &= 0x0000ffffu;
    v = (v ^ (v << 8)) & 0x00ff00ffu;
    v = (v ^ (v << 4)) & 0x0f0f0f0fu;
    v = (v ^ (v << 2)) & 0x33333333u;
    v = (v ^ (v << 1)) & 0x55555555u;
    return v;
}
inline u32 spread3(u32 x) {   // morton3D :149-156
    x = (x * 0x00010001u) & 0xFF0000FFu;
    x = (x * 0x00000101u) & 0x0F00F00Fu;
    x = (x * 0x00000011u) & 0xC30C30C3u;
    x = (x * 0x00000005u) & 0x49249249u;
    return x;
}
inline u32 shl(u32 v, u32 s) { return s >= 32 ? 0u : v << s; }     // guard host UB; device shifts use s & 31 but s<32 on every defined path
inline u32 shr(u32 v, u32 s) { return s >= 32 ? 0u : v >> s; }

u32 morton_encode(const MortonPlan& m, const float pos[3]) {        // :277-358
    int bx = m.bits[0], by = m.bits[1], bz = m.bits[2];
    const int px = m.pre[0], py = m.pre[1];
    // :281-283  min(u32(max(p * (1u << bits), 0.0f)), (1u << bits) - 1)
    u32 q0 = std::min(sat_f2u(fmaxf(pos[m.axis[0]] * (float)shl(1u, (u32)bx), 0.0f)), shl(1u, (u32)bx) - 1u);
    u32 q1 = std::min(sat_f2u(fmaxf(pos[m.axis[1]] * (float)shl(1u, (u32)by), 0.0f)), shl(1u, (u32)by) - 1u);
    u32 q2 = std::min(sat_f2u(fmaxf(pos[m.axis[2]] * (float)shl(1u, (u32)bz), 0.0f)), shl(1u, (u32)bz) - 1u);
    u32 code = 0, d0 = 0, d1 = 0;
    if (m.pre_sum > 0) {                                            // :289-338
        bx -= px;
        code = shr(q0 & shl(shl(1u, (u32)px) - 1u, (u32)bx), (u32)bx);
        code = shl(code, (u32)(py * 2));
        bx -= py; by -= py;
        u32 t0 = spread2(shr(q0 & shl(shl(1u, (u32)py) - 1u, (u32)bx), (u32)bx));
        u32 t1 = spread2(shr(q1 & shl(shl(1u, (u32)py) - 1u, (u32)by), (u32)by));
        code |= t0 * 2 + t1;
        if (m.swap > 0) {
            code <<= 1; bx -= 1;
            code |= shr(q0 & shl(1u, (u32)bx), (u32)bx);
        }
        code = shl(code, (u32)(bx + by + bz));
        q0 &= shl(1u, (u32)bx) - 1u;
        q1 &= shl(1u, (u32)by) - 1u;
        if (m.swap > 0) { d0 = (u32)(by - bx); q0 = shl(q0, d0); d1 = (u32)(by - bz); q2 = shl(q2, d1); }
        else            { d0 = (u32)(bx - by); q1 = shl(q1, d0); d1 = (u32)(bx - bz); q2 = shl(q2, d1); }
    }
    if (bz == 0) {                                                  // :340-345
        code |= spread2(q0) * 2 + spread2(q1);
    } else {                                                        // :346-356
        const u32 X = spread3(q0), Y = spread3(q1), Z = spread3(q2);
        code |= shr((m.swap > 0) ? (Y * 4 + X * 2 + Z) : (X * 4 + Y * 2 + Z), d0 + d1);
    }
    return code;
}

// 64-bit flavour of morton_encode for bit budgets up to 60 (no reference counterpart; the same steps
// with 64-bit intermediates — with NB = 30 it must reproduce morton_encode bit for bit, which tests pin).
inline u64 shl64(u64 v, u32 s) { return s >= 64 ? 0ull : v << s; }
inline u64 shr64(u64 v, u32 s) { return s >= 64 ? 0ull : v >> s; }
inline u64 spread2_64(u64 v) {
    v &= 0x00000000ffffffffull; v = (v ^ (v << 16)) & 0x0000ffff0000ffffull; v = (v ^ (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v ^ (v << 4)) & 0x0f0f0f0f0f0f0f0full; v = (v ^ (v << 2)) & 0x3333333333333333ull; v = (v ^ (v << 1)) & 0x5555555555555555ull; return v;
}
inline u64 spread3_64(u64 x) {
    x &= 0x1fffffull; x = (x | (x << 32)) & 0x1f00000000ffffull; x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full; x = (x | (x << 4)) & 0x10c30c30c30c30c3ull; x = (x | (x << 2)) & 0x1249249249249249ull; return x;
}
inline u64 sat_f2u64(float f) {
    if (f != f) return 0ull;
    if (f <= 0.0f) return 0ull;
    if (f >= 18446744073709551616.0f) return ~0ull;
    return (u64)f;
}
inline u64 quantise64(float p, int bits) {
    const u64 top = shl64(1ull, (u32)bits);
    const u64 q = sat_f2u64(fmaxf(p * (float)top, 0.0f));
    return std::min<u64>(q, top - 1ull);
}
u64 morton_encode64(const MortonPlan& m, const float pos[3]) {
    int bx = m.bits[0], by = m.bits[1], bz = m.bits[2];
    const int px = m.pre[0], py = m.pre[1];
    u64 q0 = quantise64(pos[m.axis[0]], bx), q1 = quantise64(pos[m.axis[1]], by), q2 = quantise64(pos[m.axis[2]], bz);
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

// ------------------------------------------------------------------------------------------------
// delta functions used by the hierarchy emitters
// ------------------------------------------------------------------------------------------------
inline u64 aug(const u32* k, int i) { return ((u64)k[i] << 32) | (u32)i; }
inline int clz32(u32 v) { return v ? __builtin_clz(v) : 32; }
inline int clz64(u64 v) { return v ? __builtin_clzll(v) : 64; }

// src/TwoPassLbvhKernel.h:27-40 as used at :52-54: equal keys -> 64-bit augmented clz, else clz of key xor
inline int delta2p(const u32* k, u32 i, u32 j) {
    return (k[i] == k[j]) ? clz64(aug(k, (int)i) ^ aug(k, (int)j)) : clz32(k[i] ^ k[j]);
}

// ---- the same two questions for u64 keys (60-bit codes): the augmented key {key, position} has 96 bits
struct Dist96 { u64 hi; u32 lo; };                                          // xor of two augmented keys
inline bool operator<(const Dist96& a, const Dist96& b) { return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; }
inline u64 xdist(const u32* k, int i, int j) { return aug(k, i) ^ aug(k, j); }
inline Dist96 xdist(const u64* k, int i, int j) { return { k[i] ^ k[j], (u32)i ^ (u32)j }; }
inline u64 far_dist(const u32*) { return ~0ull; }
inline Dist96 far_dist(const u64*) { return { ~0ull, ~0u }; }
inline int delta2p(const u64* k, u32 i, u32 j) { return (k[i] == k[j]) ? 64 + clz32(i ^ j) : clz64(k[i] ^ k[j]); }

// canonical (numbering-independent) topology hash: leaf -> mix(prim), internal -> mix(h(left), h(right)) (ordered)
inline u64 mix64(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x; }
inline u64 hleaf(u32 prim) { return mix64(0x9e3779b97f4a7c15ull ^ prim); }
inline u64 hnode(u64 a, u64 b) { return mix64(a * 0x100000001b3ull + mix64(b ^ 0xd6e8feb86659fd93ull)); }

} // namespace

extern "C" {

// ---- Stage E: per-primitive AABB + scene extent.  src/CommonBlocksKernel.h:92-114 (min/max => order independent)
void orc_prim_bounds(const void* tris, u32 n, void* boxes_out, void* scene_out) {
    const Tri* t = (const Tri*)tris; Box* out = (Box*)boxes_out; Box s = box_empty();
    for (u32 i = 0; i < n; ++i) { Box b = tri_box(t[i]); if (out) out[i] = b; s = unite(s, b); }
    *(Box*)scene_out = s;
}

// ---- host-built PrimRefs of the LBVH paths: src/Utility.cpp:456-477 with saMax = FltMax (identity)
void orc_primrefs(const void* tris, u32 n, void* refs_out) {
    const Tri* t = (const Tri*)tris; Leaf* out = (Leaf*)refs_out;
    for (u32 i = 0; i < n; ++i) { out[i].prim = i; out[i].b = tri_box(t[i]); }
}

// ---- Stage M.  plan_out: 9 ints {axis[3], bits[3], pre[2], pre_sum} + swap = 10 ints
void orc_morton_plan(const void* scene, int* plan_out) {
    const Box* s = (const Box*)scene;
    F3 e = { s->hi.x - s->lo.x, s->hi.y - s->lo.y, s->hi.z - s->lo.z };
    MortonPlan m = morton_plan(e);
    for (int i = 0; i < 3; ++i) { plan_out[i] = m.axis[i]; plan_out[3 + i] = m.bits[i]; }
    plan_out[6] = m.pre[0]; plan_out[7] = m.pre[1]; plan_out[8] = m.pre_sum; plan_out[9] = m.swap;
}

// src/CommonBlocksKernel.h:374-398: centre = (max+min)*0.5f; p = (centre - scene.min) / extent (f32 divides); value = index
void orc_morton_codes(const void* boxes, u32 stride_bytes, u32 box_offset_bytes, u32 n, const void* scene, u32* keys_out, u32* vals_out) {
    const Box* s = (const Box*)scene;
    const F3 e = { s->hi.x - s->lo.x, s->hi.y - s->lo.y, s->hi.z - s->lo.z };
    const MortonPlan m = morton_plan(e);
    const char* base = (const char*)boxes + box_offset_bytes;
    for (u32 i = 0; i < n; ++i) {
        Box b; std::memcpy(&b, base + (size_t)i * stride_bytes, sizeof(Box));
        const F3 c = { (b.hi.x + b.lo.x) * 0.5f, (b.hi.y + b.lo.y) * 0.5f, (b.hi.z + b.lo.z) * 0.5f };
        const float p[3] = { (c.x - s->lo.x) / e.x, (c.y - s->lo.y) / e.y, (c.z - s->lo.z) / e.z };
        keys_out[i] = morton_encode(m, p);
        if (vals_out) vals_out[i] = i;
    }
}

// the same with the per-scene plan GIVEN (int[10] as orc_morton_plan writes it) instead of derived from the extent: tests/test_gpu_round4.py feeds the plan the
// DEVICE evaluated (bvh_stage_morton_plan) to show that a key difference at an extent ratio within an ulp of a power of two is the log2f truncation and nothing else
void orc_morton_codes_plan(const void* boxes, u32 stride_bytes, u32 box_offset_bytes, u32 n, const void* scene, const int* plan, u32* keys_out) {
    const Box* s = (const Box*)scene;
    const F3 e = { s->hi.x - s->lo.x, s->hi.y - s->lo.y, s->hi.z - s->lo.z };
    MortonPlan m;
    for (int i = 0; i < 3; ++i) { m.axis[i] = plan[i]; m.bits[i] = plan[3 + i]; }
    m.pre[0] = plan[6]; m.pre[1] = plan[7]; m.pre_sum = plan[8]; m.swap = plan[9];
    const char* base = (const char*)boxes + box_offset_bytes;
    for (u32 i = 0; i < n; ++i) {
        Box b; std::memcpy(&b, base + (size_t)i * stride_bytes, sizeof(Box));
        const F3 c = { (b.hi.x + b.lo.x) * 0.5f, (b.hi.y + b.lo.y) * 0.5f, (b.hi.z + b.lo.z) * 0.5f };
        const float p[3] = { (c.x - s->lo.x) / e.x, (c.y - s->lo.y) / e.y, (c.z - s->lo.z) / e.z };
        keys_out[i] = morton_encode(m, p);
    }
}

// u64 keys with a total_bits budget (30 reproduces orc_morton_codes)
void orc_morton_codes64(const void* boxes, u32 stride_bytes, u32 box_offset_bytes, u32 n, const void* scene, int total_bits, u64* keys_out) {
    const Box* s = (const Box*)scene;
    const F3 e = { s->hi.x - s->lo.x, s->hi.y - s->lo.y, s->hi.z - s->lo.z };
    const MortonPlan m = morton_plan(e, (u32)total_bits);
    const char* base = (const char*)boxes + box_offset_bytes;
    for (u32 i = 0; i < n; ++i) {
        Box b; std::memcpy(&b, base + (size_t)i * stride_bytes, sizeof(Box));
        const F3 c = { (b.hi.x + b.lo.x) * 0.5f, (b.hi.y + b.lo.y) * 0.5f, (b.hi.z + b.lo.z) * 0.5f };
        const float p[3] = { (c.x - s->lo.x) / e.x, (c.y - s->lo.y) / e.y, (c.z - s->lo.z) / e.z };
        keys_out[i] = morton_encode64(m, p);
    }
}

// ---- Stage S: contract adopted for Oro::RadixSort::sort(src,dst,n,0,32) (call sites src/TwoPassLbvh.cpp:71-89 ...):
// stable ascending on the full 32-bit key.  parity unpinned on the reference side (implementation not in tree).
void orc_sort_pairs(const u32* keys, const u32* vals, u32 n, u32* skeys, u32* svals) {
    std::vector<u32> idx(n);
    for (u32 i = 0; i < n; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](u32 a, u32 b) { return keys[a] < keys[b]; });
    for (u32 i = 0; i < n; ++i) { skeys[i] = keys[idx[i]]; svals[i] = vals ? vals[idx[i]] : idx[i]; }
}

// ---- B-1p: Apetrei single-pass LBVH.  src/SinglePassLbvhKernel.h:27-126.  nodes_out: Node2[2n-1]; returns root index.
// Serial execution of the per-leaf walkers is one legal schedule of the kernel; the second-arriver rule makes the
// result schedule independent.
}  // extern "C"
template <typename K>
static u32 lbvh_single_impl(const void* tris, u32 n, const K* skeys, const u32* svals, void* nodes_out) {
    const Tri* t = (const Tri*)tris; Node2* nd = (Node2*)nodes_out;
    const u32 ni = n - 1;
    for (u32 g = 0; g < n; ++g) {                              // InitBvhNodes :27-54
        nd[ni + g].l = svals[g]; nd[ni + g].r = INV; nd[ni + g].b = tri_box(t[svals[g]]);
        if (g < ni) { nd[g].l = INV; nd[g].r = INV; nd[g].b = box_empty(); }
    }
    if (n == 1) return 0;   // kernel: findParent(0,1,1) == INVALID; counter[INVALID] UB in the reference; define root = the leaf
    std::vector<int> counter(n, 0);
    std::vector<u32> span_lo(n, 0), span_hi(n, 0);
    auto hdb = [&](int i, int j) {                            // findHighestDiffBit :56-62 (u32 keys: xor of the 64-bit augmented keys)
        if (j < 0 || j >= (int)n) return far_dist(skeys);
        return xdist(skeys, i, j);
    };
    auto find_parent = [&](u32 cur, int i, int j) -> u32 {    // findParent :64-86
        if (i == 0 && j == (int)n) return INV;
        if (i == 0 || (j != (int)n && hdb(j - 1, j) < hdb(i - 1, i))) { nd[j - 1].l = cur; span_lo[j - 1] = (u32)i; return (u32)(j - 1); }
        nd[i - 1].r = cur; span_hi[i - 1] = (u32)j; return (u32)(i - 1);
    };
    u32 root = INV;
    for (u32 g = 0; g < n; ++g) {                              // BvhBuildAndFit :88-126
        u32 p = find_parent(ni + g, (int)g, (int)g + 1);
        while (counter[p]++ > 0) {
            nd[p].b = unite(nd[nd[p].l].b, nd[nd[p].r].b);
            const u32 q = find_parent(p, (int)span_lo[p], (int)span_hi[p]);
            if (q == INV) { root = p; break; }
            p = q;
        }
    }
    return root;
}
extern "C" {
u32 orc_lbvh_single(const void* tris, u32 n, const u32* skeys, const u32* svals, void* nodes_out) { return lbvh_single_impl<u32>(tris, n, skeys, svals, nodes_out); }
u32 orc_lbvh_single64(const void* tris, u32 n, const u64* skeys, const u32* svals, void* nodes_out) { return lbvh_single_impl<u64>(tris, n, skeys, svals, nodes_out); }

// ---- B-2p: Karras two-pass LBVH.  src/TwoPassLbvhKernel.h:42-130,164-235.  refs: Leaf[n] (PrimRef); nodes_out Node2[2n-1];
// parents_out (optional) u32[2n-1].  Root is node 0.
}  // extern "C"
template <typename K>
static void lbvh_two_impl(const void* refs, u32 n, const K* k, const u32* svals, void* nodes_out, u32* parents_out) {
    const Leaf* pr = (const Leaf*)refs; Node2* nd = (Node2*)nodes_out;
    const u32 ni = n - 1;
    std::vector<u32> parent(2 * (size_t)n - 1, INV);
    for (u32 g = 0; g < n; ++g) {                              // InitBvhNodesPrimRef :164-194
        nd[ni + g].l = pr[svals[g]].prim; nd[ni + g].r = INV; nd[ni + g].b = pr[svals[g]].b;
        if (g < ni) { nd[g].l = INV; nd[g].r = INV; nd[g].b = box_empty(); }
    }
    for (u32 idx = 0; idx < ni; ++idx) {                       // BvhBuild :196-216
        u32 first, last;
        if (idx == 0) { first = 0; last = n - 1; }             // determineRange :44-47
        else {
            const int ld = delta2p(k, idx, idx - 1), rd = delta2p(k, idx, idx + 1);    // :52-54
            const int d = (rd > ld) ? 1 : -1;
            const int dmin_ = (ld < rd) ? ld : rd;
            auto probe = [&](long long j) -> int { return (j >= 0 && j < (long long)n) ? delta2p(k, idx, (u32)j) : -1; };
            long long lmax = 2;                                // :56-74
            while (probe((long long)idx + d * lmax) > dmin_) lmax <<= 1;
            long long l = 0;                                   // :76-91
            for (long long tt = lmax >> 1; tt > 0; tt >>= 1)
                if (probe((long long)idx + (l + tt) * d) > dmin_) l += tt;
            const u32 j = (u32)((long long)idx + l * d);
            if (d < 0) { first = j; last = idx; } else { first = idx; last = j; }
        }
        // findSplit :102-130
        const u32 dnode = (u32)delta2p(k, first, last);
        int split = (int)first, stride = (int)(last - first);
        do {
            stride = (stride + 1) >> 1;
            const int mid = split + stride;
            if ((u32)mid < last) { if ((u32)delta2p(k, first, (u32)mid) > dnode) split = mid; }
        } while (stride > 1);
        const u32 s = (u32)split;
        const u32 lc = (s == first) ? s + ni : s;              // :210-211
        const u32 rc = (s + 1 == last) ? s + 1 + ni : s + 1;
        nd[idx].l = lc; nd[idx].r = rc; parent[lc] = idx; parent[rc] = idx;
    }
    std::vector<u32> flags(2 * (size_t)n - 1, 0);              // FitBvhNodes :217-235
    if (n > 1)
        for (u32 g = 0; g < n; ++g) {
            u32 p = parent[ni + g];
            while (flags[p]++ > 0) {
                nd[p].b = unite(nd[nd[p].l].b, nd[nd[p].r].b);
                p = parent[p];
                if (p == INV) break;
            }
        }
    if (parents_out) std::memcpy(parents_out, parent.data(), parent.size() * 4);
}
extern "C" {
void orc_lbvh_two(const void* refs, u32 n, const u32* k, const u32* svals, void* nodes_out, u32* parents_out) { lbvh_two_impl<u32>(refs, n, k, svals, nodes_out, parents_out); }
void orc_lbvh_two64(const void* refs, u32 n, const u64* k, const u32* svals, void* nodes_out, u32* parents_out) { lbvh_two_impl<u64>(refs, n, k, svals, nodes_out, parents_out); }

// ---- statistics block shared by the PLOC-family emitters (feeds DESIGN.md's algorithmic-bytes figures)
struct OrcStats { u64 iterations, cluster_loads, cluster_stores, merge_calls, nn_rounds; };

// ---- B-ploc: PLOC++.  src/Ploc++Kernel.h:39-55 (SetupClusters), :211-362 (Ploc), :98-209 (SinglePassPloc),
// host loop src/PLOC++Bvh.cpp:132-152.  boxes: Box[n] by original prim index.  nodes_out: Node2[n-1]; leaves_out: Leaf[n].
// New-node numbering: the reference uses arrival order of a global counter (:57-68,:311); this restatement uses list-position
// order (one legal arrival order).  Topology is numbering independent.
void orc_ploc(const void* boxes, u32 n, const u32* svals, void* nodes_out, void* leaves_out, void* stats_out) {
    const Box* pb = (const Box*)boxes; Node2* nd = (Node2*)nodes_out; Leaf* lf = (Leaf*)leaves_out;
    const u32 ni = n - 1;
    OrcStats st = {0, 0, 0, 0, 0};
    std::vector<u32> id(n), nxt(n);
    for (u32 g = 0; g < n; ++g) {                              // SetupClusters :39-55
        lf[g].prim = svals[g]; lf[g].b = pb[svals[g]]; id[g] = g + ni;
        if (g < ni) { nd[g].l = INV; nd[g].r = INV; nd[g].b = box_empty(); }
    }
    const Box huge = { {-FLT_MAX_, -FLT_MAX_, -FLT_MAX_}, {FLT_MAX_, FLT_MAX_, FLT_MAX_} };   // :243-244
    auto box_of = [&](u32 c) -> Box { return c >= ni ? lf[c - ni].b : nd[c].b; };
    constexpr int BLK = 1024, RAD = 8, HALO = 2 * RAD;        // src/Common.h:593,595
    u32 C = n;
    std::vector<Box> cb(BLK + 2 * HALO); std::vector<u64> nn(BLK + 2 * HALO); std::vector<u32> ci(BLK + 2 * HALO);
    while (C > 1) {
        st.iterations++;
        if (C < (u32)BLK) {
            // SinglePassPloc :98-209 — whole list in one block, repeat until one cluster
            std::vector<Box> b(C); std::vector<u32> c(id.begin(), id.begin() + C);
            for (u32 i = 0; i < C; ++i) b[i] = box_of(c[i]);
            st.cluster_loads += C;
            while (C > 1) {
                std::vector<u64> m(C, ~0ull);
                for (u32 i = 0; i < C; ++i)                    // :131-148 neighbour range clipped to min(C, i+9)
                    for (u32 j = i + 1; j < std::min<u32>(C, i + RAD + 1); ++j) {
                        const u64 a = (u64)fbits(area(unite(b[j], b[i]))) << 32;
                        m[i] = std::min(m[i], a | j); m[j] = std::min(m[j], a | i);
                    }
                u32 merged = 0, out = 0;
                std::vector<Box> b2(C); std::vector<u32> c2(C);
                for (u32 i = 0; i < C; ++i) {                  // :153-184
                    const u32 nb = (u32)m[i];
                    const bool mutual = ((u32)m[nb] == i);
                    if (mutual && i < nb) {
                        const u32 node = C - 2 - merged++;
                        nd[node].l = c[i]; nd[node].r = c[nb]; nd[node].b = unite(b[i], b[nb]);
                        c2[out] = node; b2[out] = nd[node].b; ++out;
                    } else if (!mutual) { c2[out] = c[i]; b2[out] = b[i]; ++out; }
                }
                st.nn_rounds++;
                c.swap(c2); b.swap(b2); C = out;
            }
            break;
        }
        // Ploc :211-362 — chunks of BLK clusters with a 2*RAD halo either side; NN evaluated per chunk copy
        u32 merged_total = 0, out = 0;
        for (u32 o = 0; o < C; o += BLK) {
            for (int s = -HALO; s < BLK + HALO; ++s) {         // :232-249
                const long long g = (long long)o + s;
                if (g >= 0 && g < (long long)C) { ci[s + HALO] = id[g]; cb[s + HALO] = box_of(id[g]); st.cluster_loads++; }
                else { ci[s + HALO] = INV; cb[s + HALO] = huge; }
                nn[s + HALO] = ~0ull;
            }
            for (int t = -HALO; t < BLK + RAD; ++t)            // :252-270
                for (int j = t + 1; j < t + RAD + 1; ++j) {
                    const u64 a = (u64)(int64_t)(int32_t)fbits(area(unite(cb[j + HALO], cb[t + HALO]))) << 32;
                    nn[t + HALO] = std::min(nn[t + HALO], a | (u64)(int64_t)(j + (int)o));   // u64(int): sign-extends for the halo before the list
                    nn[j + HALO] = std::min(nn[j + HALO], a | (u64)(int64_t)(t + (int)o));
                }
            for (int t = 0; t < BLK && o + t < C; ++t) {       // :276-320
                const int nb = (int)((u32)nn[t + HALO] - o);
                const int nbnb = (int)((u32)nn[nb + HALO] - o);
                if (t == nbnb) {
                    if (t < nb) {
                        const u32 node = C - 2 - merged_total++;
                        nd[node].l = ci[t + HALO]; nd[node].r = ci[nb + HALO];
                        nd[node].b = unite(cb[t + HALO], cb[nb + HALO]);
                        nxt[out++] = node;
                    }
                } else nxt[out++] = ci[t + HALO];
            }
        }
        st.cluster_stores += out; st.nn_rounds++;
        id.swap(nxt); C -= merged_total;                       // src/PLOC++Bvh.cpp:150-151
    }
    if (stats_out) std::memcpy(stats_out, &st, sizeof st);
}

// ---- B-hploc: HPLOC.  src/HplocKernel.h:39-56 (SetupClusters), :66-81 (findParent), :83-117 (findNearestNeighbours),
// :126-190 (mergeClusters), :192-218 (load/storeIndices), :220-255 (plocMerge), :257-315 (HPloc).
// Serial schedule: leaf walkers run one after another; the atomicExch hand-off makes the tree schedule independent.
// The 32-slot work list models the reference's wave32 LDS arrays (WarpSize = 32 on this target, src/Common.h:100-106).
// Compaction is modelled as "valid lanes write to their rank, slot[count] = INVALID" — the outcome of :183-185 when the
// highest lane's store wins.
}  // extern "C"
// optional per-task trace of the merge tasks (tools/model_hploc.py replays it): {L, R, split, rounds, clusters loaded} per plocMerge call, in execution order
static std::vector<u32>* g_hploc_trace = nullptr;
template <typename K>
static void hploc_impl(const void* boxes, u32 n, const K* skeys, const u32* svals, void* nodes_out, void* leaves_out, void* stats_out) {
    const Box* pb = (const Box*)boxes; Node2* nd = (Node2*)nodes_out; Leaf* lf = (Leaf*)leaves_out;
    const u32 ni = n - 1;
    OrcStats st = {0, 0, 0, 0, 0};
    std::vector<u32> idx(n), par(n, INV);
    for (u32 g = 0; g < n; ++g) {                              // SetupClusters :39-56
        lf[g].prim = svals[g]; lf[g].b = pb[svals[g]]; idx[g] = g + ni;
        if (g < ni) { nd[g].l = INV; nd[g].r = INV; nd[g].b = box_empty(); }
    }
    constexpr int W = 32, HALF = 16, RAD = 8;
    u32 allocated = 0;                                         // *nMergedClusters
    auto hdb = [&](int i, int j) {                            // :58-64
        if (i < 0 || j >= (int)n) return far_dist(skeys);
        return xdist(skeys, i, j);
    };
    auto find_parent = [&](int i, int j) -> u32 {             // :66-81 (j inclusive; the j==n tests never fire)
        if (i == 0 && j == (int)n) return INV;
        if (i == 0 || (j != (int)n && hdb(j, j + 1) < hdb(i - 1, i))) return (u32)j;
        return (u32)(i - 1);
    };
    u32 slot[W]; Box sb[W]; u64 nn[W];
    auto ploc_merge = [&](u32 L, u32 R, u32 split, bool final_) {   // :220-255
        st.merge_calls++;
        for (int s = 0; s < W; ++s) { slot[s] = INV; sb[s] = box_empty(); }
        auto load = [&](u32 start, u32 end, u32 offset) -> u32 {    // loadIndices :192-206
            const u32 cnt = std::min<u32>(end - start, HALF);
            for (u32 s = 0; s < cnt; ++s) slot[s + offset] = idx[start + s];
            u32 valid = 0; for (int s = 0; s < W; ++s) valid += (slot[s] != INV);
            return std::min(cnt, valid - offset);
        };
        const u32 nl = load(L, split, 0);
        const u32 nr = load(split, R + 1, nl);
        u32 cnt = nl + nr;
        st.cluster_loads += cnt;
        const u32 cnt0 = cnt; u32 task_rounds = 0;
        const u32 threshold = final_ ? 1 : HALF;
        for (int s = 0; s < W; ++s)                            // :242-246
            if (slot[s] != INV) sb[s] = slot[s] >= ni ? lf[slot[s] - ni].b : nd[slot[s]].b;
        while (cnt > threshold) {
            st.nn_rounds++; ++task_rounds;
            for (int s = 0; s < W; ++s) nn[s] = ~0ull;         // findNearestNeighbours :83-117
            for (u32 s = 0; s < cnt; ++s)
                for (int r = 1; r <= RAD; ++r) {
                    const u32 j = s + r;
                    if (j < (u32)W && j < cnt) {
                        const u64 a = (u64)(int64_t)(int32_t)fbits(area(unite(sb[j], sb[s]))) << 32;
                        nn[s] = std::min(nn[s], a | j); nn[j] = std::min(nn[j], a | s);
                    }
                }
            u32 nslot[W]; Box nsb[W]; u32 out = 0, made = 0;   // mergeClusters :126-190
            u32 total = 0;
            for (u32 s = 0; s < cnt; ++s) { const u32 nb = (u32)nn[s]; if ((u32)nn[nb] == s && s < nb) ++total; }
            const u32 base = ni - allocated - total;           // :165-167
            allocated += total;
            for (u32 s = 0; s < cnt; ++s) {
                const u32 nb = (u32)nn[s];
                const bool mutual = ((u32)nn[nb] == s);
                if (mutual && s < nb) {
                    const u32 node = base + made++;
                    nd[node].l = slot[s]; nd[node].r = slot[nb]; nd[node].b = unite(sb[s], sb[nb]);
                    nslot[out] = node; nsb[out] = nd[node].b; ++out;
                } else if (!mutual) { nslot[out] = slot[s]; nsb[out] = sb[s]; ++out; }
            }
            for (int s = 0; s < W; ++s) { slot[s] = INV; }
            for (u32 s = 0; s < out; ++s) { slot[s] = nslot[s]; sb[s] = nsb[s]; }
            cnt = out;
        }
        for (u32 s = 0; s < nl + nr; ++s) idx[L + s] = slot[s];   // storeIndices :208-218
        st.cluster_stores += nl + nr;
        if (g_hploc_trace) { const u32 rec[5] = { L, R, split, task_rounds, cnt0 }; g_hploc_trace->insert(g_hploc_trace->end(), rec, rec + 5); }
    };
    for (u32 g = 0; g < n; ++g) {                              // HPloc :257-315, one walker at a time
        u32 L = g, R = g;
        bool active = true;
        while (active) {
            u32 split = INV, prev;
            if (find_parent((int)L, (int)R) == R) {            // :276-285
                prev = par[R]; par[R] = L;
                if (prev != INV) { split = R + 1; R = prev; }
            } else {                                           // :286-295
                prev = par[L - 1]; par[L - 1] = R;
                if (prev != INV) { split = L; L = prev; }
            }
            if (prev == INV) { active = false; break; }
            const u32 size = R - L + 1;
            const bool final_ = (size == n);
            if (size > (u32)HALF || final_) ploc_merge(L, R, split, final_);   // :303-312
        }
    }
    if (stats_out) std::memcpy(stats_out, &st, sizeof st);
}
extern "C" {
void orc_hploc(const void* boxes, u32 n, const u32* skeys, const u32* svals, void* nodes_out, void* leaves_out, void* stats_out) { hploc_impl<u32>(boxes, n, skeys, svals, nodes_out, leaves_out, stats_out); }
void orc_hploc64(const void* boxes, u32 n, const u64* skeys, const u32* svals, void* nodes_out, void* leaves_out, void* stats_out) { hploc_impl<u64>(boxes, n, skeys, svals, nodes_out, leaves_out, stats_out); }
// the merge tasks of an HPLOC build: tasks_out = u32[5 * cap] {L, R, split, rounds, clusters loaded}; returns the number of tasks (<= cap are written)
u32 orc_hploc_tasks(const void* boxes, u32 n, const u32* skeys, const u32* svals, u32* tasks_out, u32 cap) {
    std::vector<Node2> nodes(n > 1 ? n - 1 : 1); std::vector<Leaf> leaves(n);
    std::vector<u32> tr; g_hploc_trace = &tr;
    hploc_impl<u32>(boxes, n, skeys, svals, nodes.data(), leaves.data(), nullptr);
    g_hploc_trace = nullptr;
    const u32 cnt = (u32)(tr.size() / 5);
    std::memcpy(tasks_out, tr.data(), sizeof(u32) * 5 * std::min(cnt, cap));
    return cnt;
}

// ---- SAH cost, BVH2.  Formula of Utility::calculateLbvhCost (src/Utility.cpp:317-349): 1 + sum over internal nodes of both
// child areas / rootArea + sum over leaves of area / rootArea.  Returned in f64 (order independent to ~1e-12) and, through
// *f32_out, accumulated in f32 in node-index order exactly as the reference does.
// layout 0: LBVH (one array of 2n-1, leaves at n-1+i).  layout 1: PLOC (nodes[n-1] + leaves[n]; child >= n-1 is a leaf).
double orc_sah_bvh2(const void* nodes, const void* leaves, u32 root, u32 n, int layout, float* f32_out) {
    const Node2* nd = (const Node2*)nodes; const Leaf* lf = (const Leaf*)leaves;
    const u32 ni = n - 1;
    auto box_of = [&](u32 c) -> Box { return (layout == 1 && c >= ni) ? lf[c - ni].b : nd[c].b; };
    const float root_area = area(box_of(root));
    const float inv = 1.0f / root_area;
    double c64 = 1.0; float c32 = 1.0f;
    for (u32 i = 0; i < ni; ++i) {
        if (nd[i].l != INV) { const float a = area(box_of(nd[i].l)); c32 += 1.0f * a * inv; c64 += (double)a / (double)root_area; }
        if (nd[i].r != INV) { const float a = area(box_of(nd[i].r)); c32 += 1.0f * a * inv; c64 += (double)a / (double)root_area; }
    }
    for (u32 i = 0; i < n; ++i) {
        const bool present = (layout == 1) ? true : (nd[ni + i].l != INV);
        if (present) { const float a = area(layout == 1 ? lf[i].b : nd[ni + i].b); c32 += 1.0f * a * inv; c64 += (double)a / (double)root_area; }
    }
    if (f32_out) *f32_out = c32;
    return c64;
}

// ---- validators.  Same acceptance criteria as Utility::checkLBvhCorrectness / checkPlocBvh2Correctness
// (src/Utility.cpp:31-91): every primitive reachable exactly once from the root — but with an unbounded stack (the
// reference's 32-entry stack overflows on deep LBVH trees) — plus: every internal box equals the
// union of its children (bit exact), every internal node visited exactly once.
// returns 0 on success, otherwise a bit mask {1: prim coverage, 2: box mismatch, 4: node reuse/cycle, 8: bad index}
int orc_validate_bvh2(const void* nodes, const void* leaves, u32 root, u32 n, int layout) {
    const Node2* nd = (const Node2*)nodes; const Leaf* lf = (const Leaf*)leaves;
    const u32 ni = n - 1; int err = 0;
    if (n == 1) return 0;
    std::vector<uint8_t> seen_prim(n, 0), seen_node(ni, 0);
    std::vector<u32> stack; stack.push_back(root);
    u32 prims = 0;
    auto box_of = [&](u32 c) -> Box { return (layout == 1 && c >= ni) ? lf[c - ni].b : nd[c].b; };
    while (!stack.empty()) {
        const u32 c = stack.back(); stack.pop_back();
        if (c >= ni) {
            if (c - ni >= n) { err |= 8; continue; }
            const u32 p = (layout == 1) ? lf[c - ni].prim : nd[c].l;
            if (p >= n) { err |= 8; continue; }
            if (seen_prim[p]++) err |= 1;
            ++prims;
        } else {
            if (seen_node[c]++) { err |= 4; continue; }
            const u32 l = nd[c].l, r = nd[c].r;
            if (l == INV || r == INV || l >= 2 * (size_t)n - 1 || r >= 2 * (size_t)n - 1) { err |= 8; continue; }
            const Box u = unite(box_of(l), box_of(r));
            if (std::memcmp(&u, &nd[c].b, sizeof(Box)) != 0) err |= 2;
            stack.push_back(l); stack.push_back(r);
        }
    }
    if (prims != n) err |= 1;
    return err;
}

// canonical topology hash (numbering independent; child order significant)
u64 orc_topology_hash(const void* nodes, const void* leaves, u32 root, u32 n, int layout) {
    const Node2* nd = (const Node2*)nodes; const Leaf* lf = (const Leaf*)leaves;
    const u32 ni = n - 1;
    if (n == 1) return hleaf(layout == 1 ? lf[0].prim : nd[0].l);
    std::vector<u64> h(ni, 0); std::vector<uint8_t> state(ni, 0);
    auto hash_of = [&](u32 c) -> u64 { return c >= ni ? hleaf(layout == 1 ? lf[c - ni].prim : nd[c].l) : h[c]; };
    std::vector<u32> stack; stack.push_back(root);
    while (!stack.empty()) {
        const u32 c = stack.back();
        if (c >= ni) { stack.pop_back(); continue; }
        if (state[c] == 0) { state[c] = 1; stack.push_back(nd[c].l); stack.push_back(nd[c].r); }
        else { if (state[c] == 1) { h[c] = hnode(hash_of(nd[c].l), hash_of(nd[c].r)); state[c] = 2; } stack.pop_back(); }
    }
    return hash_of(root);
}

// FNV-1a over a byte range (array fingerprints in tests/golden)
u64 orc_fnv1a(const void* p, u64 bytes) {
    const uint8_t* b = (const uint8_t*)p; u64 h = 0xcbf29ce484222325ull;
    for (u64 i = 0; i < bytes; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}

// PLOC layout (nodes[n-1] + leaves[n], root 0) -> LBVH layout (one array of 2n-1, leaf i at n-1+i).  The adapter the
// reference never wrote: leaf record {left = primIdx, right = INVALID, aabb} as
// src/TwoPassLbvhKernel.h:177-184; internal nodes copied verbatim (child indices already follow the >= n-1 convention).
void orc_ploc_to_lbvh_layout(const void* nodes, const void* leaves, u32 n, void* out_nodes) {
    const Node2* nd = (const Node2*)nodes; const Leaf* lf = (const Leaf*)leaves; Node2* o = (Node2*)out_nodes;
    const u32 ni = n - 1;
    for (u32 i = 0; i < ni; ++i) o[i] = nd[i];
    for (u32 i = 0; i < n; ++i) { o[ni + i].l = lf[i].prim; o[ni + i].r = INV; o[ni + i].b = lf[i].b; }
}

// ---- CPU baseline: the reference's 32-bucket binned-SAH builder.  src/BinnedSahBvh.cpp:13-203 (the part before the
// ray-tracing call at :209).  Quirks kept: partition / nth_element over [start, end-1) (:75-76,:145,:165-166,:179-180);
// min search over buckets 0..30 (:136-142); leaf at 1 prim (:50-56); BFS node order, children adjacent (:66-68).
// Deviation (documented in DESIGN.md): the bucket index in the counting loop is clamped to 31 — the reference writes out of
// bounds at :101-104 when a centroid lies on the node's max plane.
// nodes_out: SahNode[3n-1].  Returns the number of nodes used.
u32 orc_binned_sah_build(const void* tris, u32 n, void* nodes_out) {
    const Tri* t = (const Tri*)tris; SahNode* nodes = (SahNode*)nodes_out;
    struct Ref { Box b; size_t prim; };
    struct Task { u32 node, start, end; };
    std::vector<Ref> refs; refs.reserve(n);
    for (u32 i = 0; i < n; ++i) refs.push_back({ tri_box(t[i]), i });   // :15-23
    constexpr u32 NBUCKET = 32;
    std::queue<Task> q; q.push({0, 0, n});
    u32 next = 0;
    nodes[next].first = 0; nodes[next].count = 0; ++next;
    auto centre = [](const Box& b, int d) { return d == 0 ? (b.hi.x + b.lo.x) * 0.5f : d == 1 ? (b.hi.y + b.lo.y) * 0.5f : (b.hi.z + b.lo.z) * 0.5f; };
    auto offset = [&](const Box& nb, const Box& b, int d) {    // Aabb::offset, Common.h:367-374
        const float c = centre(b, d);
        const float lo = d == 0 ? nb.lo.x : d == 1 ? nb.lo.y : nb.lo.z, hi = d == 0 ? nb.hi.x : d == 1 ? nb.hi.y : nb.hi.z;
        float o = c - lo; if (hi > lo) o /= hi - lo; return o;
    };
    while (!q.empty()) {
        const Task tk = q.front(); q.pop();
        SahNode& node = nodes[tk.node];
        if (tk.end - tk.start == 1) {                          // :50-56
            node.b = refs[tk.start].b; node.first = (u32)refs[tk.start].prim; node.count = 1; continue;
        }
        Box nb = box_empty();
        for (u32 i = tk.start; i < tk.end; ++i) nb = unite(nb, refs[i].b);
        node.b = nb;
        const float ex = nb.hi.x - nb.lo.x, ey = nb.hi.y - nb.lo.y, ez = nb.hi.z - nb.lo.z;
        const int dim = (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);   // maximumExtentDim, Common.h:351-359
        node.first = next++; node.count = 0; ++next;
        const u32 first_child = node.first;
        u32 split = 0;
        auto by_centre = [&](const Ref& a, const Ref& b) { return centre(a.b, dim) < centre(b.b, dim); };
        if (tk.end - tk.start <= 2) {                          // :72-91
            split = (tk.start + tk.end) / 2;
            std::nth_element(&refs[tk.start], &refs[split], &refs[tk.end - 1], by_centre);
        } else {
            struct Bucket { int cnt = 0; Box b = box_empty(); } bk[NBUCKET];
            for (u32 i = tk.start; i < tk.end; ++i) {          // :95-105
                u32 b = (u32)(NBUCKET * offset(nb, refs[i].b, dim));
                if (b >= NBUCKET) b = NBUCKET - 1;              // deviation: clamp (reference: OOB write)
                bk[b].cnt++; bk[b].b = unite(bk[b].b, refs[i].b);
            }
            float cost[NBUCKET];
            for (u32 b = 0; b < NBUCKET; ++b) {                // :107-131
                Box lh = box_empty(), rh = box_empty(); int lc = 0, rc = 0;
                for (u32 j = 0; j <= b; ++j) { if (!bk[j].cnt) continue; lh = unite(lh, bk[j].b); lc += bk[j].cnt; }
                for (u32 j = b + 1; j < NBUCKET; ++j) { if (!bk[j].cnt) continue; rh = unite(rh, bk[j].b); rc += bk[j].cnt; }
                const float ls = lc == 0 ? 0.0f : lc * area(lh), rs = rc == 0 ? 0.0f : rc * area(rh);
                const float tot = (lc + rc) == 0 ? 0.0f : ((ls + rs) / area(nb));
                cost[b] = (tot == 0.0f) ? FLT_MAX_ : 0.125f + tot;
            }
            float best = cost[0]; int sb = 0;                  // :133-142
            for (u32 i = 0; i < NBUCKET - 1; ++i) if (cost[i] < best) { best = cost[i]; sb = (int)i; }
            split = (u32)(std::partition(&refs[tk.start], &refs[tk.end - 1], [&](const Ref& r) {   // :144-153
                u32 b = (u32)(NBUCKET * offset(nb, r.b, dim)); if (b == NBUCKET) b = NBUCKET - 1;
                return (int)b <= sb; }) - &refs[0]);
            if (split <= tk.start || split >= tk.end) {        // :156-173
                const float mid = offset(nb, nb, dim) / 2.0f;
                split = (u32)(std::partition(&refs[tk.start], &refs[tk.end - 1], [&](const Ref& r) { return centre(r.b, dim) < mid; }) - &refs[0]);
            }
            if (split <= tk.start || split >= tk.end) {        // :176-195
                split = (tk.start + tk.end) / 2;
                std::nth_element(&refs[tk.start], &refs[split], &refs[tk.end - 1], by_centre);
            }
        }
        q.push({ first_child, tk.start, split });              // :198-202
        q.push({ first_child + 1, split, tk.end });
    }
    return next;
}

// correct SAH cost of a binned-SAH tree (f64) with the calculateLbvhCost formula; *ref_formula_out receives the value of the
// reference's calculateBinnedSahBvhCost (src/Utility.cpp:398-422), which reads a leaf's primId as a child index.
double orc_sah_binned(const void* nodes_, u32 total, u32 n, float* ref_formula_out) {
    const SahNode* nd = (const SahNode*)nodes_;
    const float ra = area(nd[0].b); double c = 1.0;
    for (u32 i = 0; i < total; ++i) {
        if (nd[i].count == 0) { c += (double)area(nd[nd[i].first].b) / ra + (double)area(nd[nd[i].first + 1].b) / ra; }
        else c += (double)area(nd[i].b) / ra;
    }
    if (ref_formula_out) {
        const float inv = 1.0f / ra; float cost = 1.0f;
        const u32 cap = 3 * n - 1;
        for (u32 i = 0; i < total; ++i) {
            if (nd[i].first != INV && nd[i].first < cap) cost += 1.0f * area(nd[nd[i].first].b) * inv;
            if (nd[i].first + 1 != INV && nd[i].first + 1 < cap) cost += 1.0f * area(nd[nd[i].first + 1].b) * inv;
        }
        *ref_formula_out = cost;
    }
    return c;
}

// ---- BVH2 -> BVH4 collapse.  src/TwoPassLbvhKernel.h:256-331 semantics as a BFS (the GPU kernel's spin-wait scheduling
// only changes wide-node numbering).  layout as above.  bvh4_out: Node4[n], prim_out: PrimNode[n].  Returns wide node count.
u32 orc_collapse4(const void* nodes, const void* leaves, u32 root, u32 n, int layout, void* bvh4_out, void* prim_out) {
    const Node2* nd = (const Node2*)nodes; const Leaf* lf = (const Leaf*)leaves;
    Node4* w = (Node4*)bvh4_out; PrimNode* pn = (PrimNode*)prim_out;
    const u32 ni = n - 1;
    auto box_of = [&](u32 c) -> Box { return (layout == 1 && c >= ni) ? lf[c - ni].b : nd[c].b; };
    auto prim_of = [&](u32 c) -> u32 { return layout == 1 ? lf[c - ni].prim : nd[c].l; };
    struct T { u32 bvh2, parent; };
    std::vector<T> task(n, T{INV, INV});
    task[0] = { root, INV };
    u32 next = 1;
    for (u32 g = 0; g < next; ++g) {
        const Node2& n2 = nd[task[g].bvh2];
        u32 ci[4] = { n2.l, n2.r, INV, INV }; Box cb[4] = { box_of(n2.l), box_of(n2.r), box_empty(), box_empty() };
        u32 cc = 2;
        for (int j = 0; j < 2; ++j) {                          // :270-296
            float best = 0.0f; u32 pos = INV;
            for (u32 k = 0; k < cc; ++k) if (ci[k] < ni) { const float a = area(nd[ci[k]].b); if (a > best) { pos = k; best = a; } }
            if (pos == INV) break;
            const Node2 mc = nd[ci[pos]];
            ci[pos] = mc.l; cb[pos] = box_of(mc.l); ci[cc] = mc.r; cb[cc] = box_of(mc.r); ++cc;
        }
        Node4 wn; for (int k = 0; k < 4; ++k) { wn.b[k] = box_empty(); wn.child[k] = INV; }
        wn.parent = task[g].parent; wn.count = cc;
        for (u32 k = 0; k < cc; ++k) {                         // :309-325
            if (ci[k] < ni) { wn.child[k] = next; wn.b[k] = cb[k]; task[next] = { ci[k], g }; ++next; }
            else { wn.child[k] = ci[k]; pn[ci[k] - ni].parent = g; pn[ci[k] - ni].prim = prim_of(ci[k]); }
        }
        w[g] = wn;
    }
    return next;
}

// canonical (numbering independent, child-slot order significant) hash of a BVH4; also checks structure:
// returns 0 if a wide node is visited twice, a child index is out of range, or a PrimNode's parent link is wrong
u64 orc_topology_hash4(const void* bvh4, const void* prim_nodes, u32 total, u32 n) {
    const Node4* w = (const Node4*)bvh4; const PrimNode* pn = (const PrimNode*)prim_nodes;
    const u32 ni = n - 1;
    std::vector<u64> h(total, 0); std::vector<uint8_t> state(total, 0);
    std::vector<u32> stack; stack.push_back(0);
    while (!stack.empty()) {
        const u32 c = stack.back();
        if (c >= total) return 0;
        if (state[c] == 0) {
            state[c] = 1;
            if (w[c].count < 2 || w[c].count > 4) return 0;
            for (u32 k = 0; k < w[c].count; ++k) {
                const u32 ch = w[c].child[k];
                if (ch < ni) { if (ch >= total || state[ch] != 0 || w[ch].parent != c) return 0; stack.push_back(ch); }
                else if (ch - ni >= n || pn[ch - ni].parent != c) return 0;
            }
        } else {
            if (state[c] == 1) {
                u64 acc = 0x243f6a8885a308d3ull + w[c].count;
                for (u32 k = 0; k < w[c].count; ++k) {
                    const u32 ch = w[c].child[k];
                    acc = hnode(acc, ch < ni ? h[ch] : hleaf(pn[ch - ni].prim));
                }
                h[c] = acc; state[c] = 2;
            }
            stack.pop_back();
        }
    }
    return h[0];
}

// Utility::calculatebvh4Cost, src/Utility.cpp:351-396 (f32, node-index order) + f64 twin
double orc_sah_bvh4(const void* bvh4, const void* prim_nodes, const void* prim_boxes, u32 total, u32 n, float* f32_out) {
    const Node4* w = (const Node4*)bvh4; const PrimNode* pn = (const PrimNode*)prim_nodes; const Box* pb = (const Box*)prim_boxes;
    const u32 ni = n - 1;
    Box rb = box_empty();
    for (int k = 0; k < 4; ++k) if (w[0].child[k] != INV) rb = unite(rb, w[0].b[k]);
    const float ra = area(rb), inv = 1.0f / ra;
    float c32 = 1.0f; double c64 = 1.0;
    for (u32 i = 0; i < total; ++i)
        for (int k = 0; k < 4; ++k)
            if (w[i].child[k] != INV && w[i].child[k] < ni) { const float a = area(w[i].b[k]); c32 += 1.0f * a * inv; c64 += (double)a / ra; }
    for (u32 i = 0; i < n; ++i) { const float a = area(pb[pn[i].prim]); c32 += a * inv; c64 += (double)a / ra; }
    if (f32_out) *f32_out = c32;
    return c64;
}

} // extern "C"
