// trace_oracle.cpp — CPU restatement of the reference's primary-ray generation and while-while BVH2 traversal (the consumer
// side of BASELINE.json config 4's image check).  TEST INFRASTRUCTURE ONLY (see bvh_oracle.cpp).
//   rays      : GenerateRays, src/CommonBlocksKernel.h:432-463
//   traversal : BvhTraversalWhile, src/TraversalKernel.h:238-335 (plain while-while; LBVH layout: leaves in the same array)
//   math      : quaternion / transform / triangle test of src/Common.h:461-531, slab test Aabb::intersect src/Common.h:384-397
// Built with -ffp-contract=off so that every operation rounds exactly as the device code built the same way.
#include <cmath>
#include <cstdint>
#include <cstring>

namespace {
using u32 = uint32_t; using u8 = uint8_t;
constexpr u32 INV = 0xFFFFFFFFu; constexpr float FMAXV = 3.402823466e+38f;
struct F3 { float x, y, z; }; struct F4 { float x, y, z, w; }; struct F2 { float x, y; };
struct Box { F3 lo, hi; };
struct alignas(64) Tri { F3 a, b, c; };
struct alignas(32) Node2 { u32 l, r; Box b; };
struct alignas(32) Ray { F3 o, d; float tmin, tmax; };                         // src/Common.h:533-539
struct alignas(64) Camera { F4 eye, quat; float fov, near_, far_, pad; };      // src/Common.h:550-558
struct alignas(64) Xform { F3 t; float p0; F3 s; float p1; F4 q; };            // src/Common.h:541-548
static_assert(sizeof(Ray) == 32 && sizeof(Camera) == 64 && sizeof(Xform) == 64, "layout");

inline F3 operator+(F3 a, F3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline F3 operator-(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline F3 operator*(F3 a, F3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline F3 operator*(float c, F3 a) { return { c * a.x, c * a.y, c * a.z }; }
inline F3 operator/(F3 a, F3 b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
inline F3 operator/(F3 a, float b) { return { a.x / b, a.y / b, a.z / b }; }
inline F3 rdiv(float b, F3 a) { return { b / a.x, b / a.y, b / a.z }; }
inline F4 operator+(F4 a, F4 b) { return { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
inline float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline F3 cross(F3 a, F3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline F3 normalize(F3 a) { return a / sqrtf(dot(a, a)); }
inline F4 qmul(F4 a, F4 b) {                                                    // qtMul :483-492
    const F3 c = cross({ a.x, a.y, a.z }, { b.x, b.y, b.z });
    F4 r = { c.x, c.y, c.z, 0.0f };
    r = r + F4{ a.w * b.x, a.w * b.y, a.w * b.z, a.w * b.w } + F4{ b.w * a.x, b.w * a.y, b.w * a.z, b.w * a.w };
    r.w = a.w * b.w - dot({ a.x, a.y, a.z }, { b.x, b.y, b.z });
    return r;
}
inline F4 qinv(F4 q) { return { -q.x, -q.y, -q.z, q.w }; }
inline F3 qrot(F4 q, F3 p) { const F4 o = qmul(qmul(q, { p.x, p.y, p.z, 0.0f }), qinv(q)); return { o.x, o.y, o.z }; }   // :502-508
inline F3 inv_transform(F3 p, F3 s, F4 r, F3 t) { return qrot(qinv(r), p - t) / s; }                                      // :510-512
inline F3 transform(F3 p, F3 s, F4 r, F3 t) { return qrot(r, s * p) + t; }                                                // :514
inline F2 slab(const Box& b, F3 from, F3 inv, float maxt) {                                                                // :384-397
    const F3 df = (b.hi - from) * inv, dn = (b.lo - from) * inv;
    const F3 tf = { fmaxf(df.x, dn.x), fmaxf(df.y, dn.y), fmaxf(df.z, dn.z) }, tn = { fminf(df.x, dn.x), fminf(df.y, dn.y), fminf(df.z, dn.z) };
    float mf = fminf(tf.x, fminf(tf.y, tf.z)), mn = fmaxf(tn.x, fmaxf(tn.y, tn.z));
    mf = fminf(maxt, mf); mn = fmaxf(0.0f, mn);
    return { mn, mf };
}
inline F4 tri_hit(F3 v0, F3 v1, F3 v2, F3 o, F3 d) {                                                                        // :516-531
    const F3 p0 = v0 - o, p1 = v1 - o, p2 = v2 - o, e0 = v2 - v0, e1 = v0 - v1, e2 = v1 - v2;
    const F3 nrm = cross(e1, e0);
    const float u = dot(cross(p0 + p2, e0), d), v = dot(cross(p1 + p0, e1), d), w = dot(cross(p2 + p1, e2), d), t = dot(p0, nrm) * 2.0f;
    const float den = dot(nrm, d) * 2.0f;
    return { u / den, v / den, w / den, t / den };
}
inline u8 to_u8(float f) { if (!(f > 0.0f)) return 0; if (f >= 4294967296.0f) return 255; return (u8)(u32)f; }   // v_cvt_u32_f32 then truncate
} // namespace

extern "C" {

// GenerateRays :432-463.  rays[gx*height + gy]
void orc_generate_rays(const void* cam_, void* rays_, u32 width, u32 height) {
    const Camera& cam = *(const Camera*)cam_; Ray* rays = (Ray*)rays_;
    const float sx = 0.024f * (width / (float)height), sy = 0.024f;
    const F3 hol = qrot(cam.quat, { 1, 0, 0 }), up = qrot(cam.quat, { 0, -1, 0 }), view = qrot(cam.quat, { 0, 0, -1 });
    for (u32 gx = 0; gx < width; ++gx)
        for (u32 gy = 0; gy < height; ++gy) {
            const float px = ((float)gx + 0.5f) / width - 0.5f, py = ((float)gy + 0.5f) / height - 0.5f;
            F3 dir = { px * sx, py * sy, sy / (2.f * tanf(cam.fov / 2.f)) };
            dir = normalize(dir.x * hol + dir.y * up + dir.z * view);
            Ray& r = rays[gx * height + gy];
            r.o = { cam.eye.x, cam.eye.y, cam.eye.z };
            const F4 far4 = cam.eye + F4{ dir.x * cam.far_, dir.y * cam.far_, dir.z * cam.far_, 0.0f };
            r.d = normalize({ far4.x, far4.y, far4.z });
            r.tmin = 0.0f; r.tmax = FMAXV;
        }
}

// BvhTraversalWhile :238-335 over an LBVH-layout node array (leaf = index >= n_internal, left = primitive index).
// rgba must be zero-initialised by the caller (pixels of missing rays are left untouched, :444-450).  Stack: 64 entries and the
// reference's `top < 64` guard (:296) — the reference's LDS stack holds only 32; *overflow_out counts
// rays that needed more than 32.
void orc_trace_while(const void* rays_, const void* tris_, const void* nodes_, const void* xf_, u8* rgba, u32 root, u32 width, u32 height,
                     u32 n_internal, u32* overflow_out) {
    const Ray* rays = (const Ray*)rays_; const Tri* tris = (const Tri*)tris_; const Node2* nodes = (const Node2*)nodes_; const Xform& tr = *(const Xform*)xf_;
    u32 overflow = 0;
    for (u32 gx = 0; gx < width; ++gx)
        for (u32 gy = 0; gy < height; ++gy) {
            const u32 index = gx * width + gy;
            const Ray ray = rays[index];
            u32 node = root, top = 0, stack[64], deepest = 0;
            stack[top++] = INV;
            u32 hit_prim = INV; float hit_t = FMAXV; F2 uv = { 0, 0 };
            const F3 o = inv_transform(ray.o, tr.s, tr.q, tr.t), d = inv_transform(ray.d, tr.s, tr.q, { 0, 0, 0 });
            const F3 inv = rdiv(1.0f, d);
            while (node != INV) {
                while (node < n_internal) {
                    const Node2& nd = nodes[node];
                    const F2 t0 = slab(nodes[nd.l].b, o, inv, hit_t), t1 = slab(nodes[nd.r].b, o, inv, hit_t);
                    const bool hl = t0.x <= t0.y, hr = t1.x <= t1.y;
                    if (hl || hr) {
                        if (hl && hr) {
                            node = (t0.x < t1.x) ? nd.l : nd.r;
                            if (top < 64) { stack[top++] = (t0.x < t1.x) ? nd.r : nd.l; if (top > deepest) deepest = top; }
                        } else node = hl ? nd.l : nd.r;
                        continue;
                    }
                    node = stack[--top];
                }
                while (node >= n_internal && node != INV) {
                    const Node2& nd = nodes[node];
                    const Tri& t = tris[nd.l];
                    const F4 it = tri_hit(transform(t.a, tr.s, tr.q, tr.t), transform(t.b, tr.s, tr.q, tr.t), transform(t.c, tr.s, tr.q, tr.t), ray.o, ray.d);
                    if (it.x > 0.0f && it.y > 0.0f && it.z > 0.0f && it.w > 0.0f && it.w < hit_t) { hit_prim = nd.l; hit_t = it.w; uv = { it.x, it.y }; }
                    node = stack[--top];
                }
            }
            if (deepest > 32) ++overflow;
            if (hit_prim != INV) {
                rgba[index * 4 + 0] = to_u8(uv.x * 255); rgba[index * 4 + 1] = to_u8(uv.y * 255);
                rgba[index * 4 + 2] = to_u8((1 - uv.x - uv.y) * 255); rgba[index * 4 + 3] = 255;
            }
        }
    if (overflow_out) *overflow_out = overflow;
}

} // extern "C"
