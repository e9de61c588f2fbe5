// trace.hip — the consumer side of the build path, needed for BASELINE.json config 4's image check:
// primary-ray generation and while-while BVH2 traversal over the LBVH node layout, producing the reference's RGBA image.
//   k_generate_rays replaces GenerateRays            (reference src/CommonBlocksKernel.h:432-463)
//   k_trace_while   replaces BvhTraversalWhile       (reference src/TraversalKernel.h:238-335)
//   k_trace<kind>   replaces BvhTraversalRestartTrail (:49-146), BvhTraversalifif (:148-236), BvhTraversalSpeculativeWhile (:337-451)
// with the math of src/Common.h:461-531 (quaternions, transforms, triangle test) and Aabb::intersect (src/Common.h:384-397).
// PLOC/HPLOC trees are traversed through bvh_to_lbvh_layout (the adapter the reference never wrote).
// Arithmetic is kept operation-for-operation (file built with -ffp-contract=off) so that images are pixel-exact against the
// CPU oracle.  The per-ray stack holds 64 entries in LDS — the reference guards with `top < 64` but reserves only 32 per
// thread; rays that would overflow 32 are the only place results can differ from the reference.
#include "common.hpp"
#include "kernels.hpp"

namespace bvh {

struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };
struct alignas(32) RayRec { F3 o, d; float tmin, tmax; };                          // Ray, src/Common.h:533-539
struct alignas(64) CameraRec { F4 eye, quat; float fov, near_, far_, pad; };       // Camera, src/Common.h:550-558
struct alignas(64) XformRec { F3 t; float p0; F3 s; float p1; F4 q; };             // Transformation, src/Common.h:541-548
static_assert(sizeof(RayRec) == 32 && sizeof(CameraRec) == 64 && sizeof(XformRec) == 64, "reference layouts");

__device__ __forceinline__ F3 add(F3 a, F3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
__device__ __forceinline__ F3 sub(F3 a, F3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
__device__ __forceinline__ F3 mul(F3 a, F3 b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
__device__ __forceinline__ F3 scale(float c, F3 a) { return { c * a.x, c * a.y, c * a.z }; }
__device__ __forceinline__ F3 div3(F3 a, F3 b) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
__device__ __forceinline__ F3 divs(F3 a, float b) { return { a.x / b, a.y / b, a.z / b }; }
__device__ __forceinline__ F4 add4(F4 a, F4 b) { return { a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w }; }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
__device__ __forceinline__ F3 normalize3(F3 a) { return divs(a, sqrtf(dot3(a, a))); }
__device__ __forceinline__ F4 qmul(F4 a, F4 b) {                                   // qtMul :483-492
    const F3 c = cross3({ a.x, a.y, a.z }, { b.x, b.y, b.z });
    F4 r = { c.x, c.y, c.z, 0.0f };
    r = add4(add4(r, { a.w * b.x, a.w * b.y, a.w * b.z, a.w * b.w }), { b.w * a.x, b.w * a.y, b.w * a.z, b.w * a.w });
    r.w = a.w * b.w - dot3({ a.x, a.y, a.z }, { b.x, b.y, b.z });
    return r;
}
__device__ __forceinline__ F4 qinv(F4 q) { return { -q.x, -q.y, -q.z, q.w }; }
__device__ __forceinline__ F3 qrot(F4 q, F3 p) { const F4 o = qmul(qmul(q, { p.x, p.y, p.z, 0.0f }), qinv(q)); return { o.x, o.y, o.z }; }
__device__ __forceinline__ F3 inv_transform(F3 p, F3 s, F4 r, F3 t) { return div3(qrot(qinv(r), sub(p, t)), s); }
__device__ __forceinline__ F3 transform(F3 p, F3 s, F4 r, F3 t) { return add(qrot(r, mul(s, p)), t); }

__global__ void k_generate_rays(const CameraRec* __restrict__ cam_, RayRec* __restrict__ rays, u32 width, u32 height) {
    const u32 gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y * blockDim.y + threadIdx.y;
    if (gx >= width || gy >= height) return;
    const CameraRec cam = *cam_;
    const float sx = 0.024f * (width / (float)height), sy = 0.024f;
    const float px = ((float)gx + 0.5f) / width - 0.5f, py = ((float)gy + 0.5f) / height - 0.5f;
    F3 dir = { px * sx, py * sy, sy / (2.f * tanf(cam.fov / 2.f)) };
    const F3 hol = qrot(cam.quat, { 1, 0, 0 }), up = qrot(cam.quat, { 0, -1, 0 }), view = qrot(cam.quat, { 0, 0, -1 });
    dir = normalize3(add(add(scale(dir.x, hol), scale(dir.y, up)), scale(dir.z, view)));
    RayRec r;
    r.o = { cam.eye.x, cam.eye.y, cam.eye.z };
    const F4 far4 = add4(cam.eye, { dir.x * cam.far_, dir.y * cam.far_, dir.z * cam.far_, 0.0f });
    r.d = normalize3({ far4.x, far4.y, far4.z });
    r.tmin = 0.0f; r.tmax = FMAX;
    rays[gx * height + gy] = r;                                                    // :457-461
}

__device__ __forceinline__ void slab(const bvh_aabb& b, F3 from, F3 inv, float maxt, float& tnear, float& tfar) {   // Aabb::intersect
    const F3 df = mul(sub({ b.max.x, b.max.y, b.max.z }, from), inv), dn = mul(sub({ b.min.x, b.min.y, b.min.z }, from), inv);
    const F3 tf = { fmaxf(df.x, dn.x), fmaxf(df.y, dn.y), fmaxf(df.z, dn.z) }, tn = { fminf(df.x, dn.x), fminf(df.y, dn.y), fminf(df.z, dn.z) };
    float mf = fminf(tf.x, fminf(tf.y, tf.z)), mn = fmaxf(tn.x, fmaxf(tn.y, tn.z));
    tfar = fminf(maxt, mf); tnear = fmaxf(0.0f, mn);
}
__device__ __forceinline__ unsigned char to_u8(float f) { return (unsigned char)(u32)fminf(fmaxf(f, 0.0f), 4294967040.0f); }

constexpr int TR_STACK = 64;
__global__ __launch_bounds__(64) void k_trace_while(const RayRec* __restrict__ rays, const bvh_triangle* __restrict__ tris,
                                                    const bvh2_node* __restrict__ nodes, const XformRec* __restrict__ xf,
                                                    unsigned char* __restrict__ rgba, u32 root, u32 width, u32 height, u32 n_internal) {
    __shared__ u32 s_stack[TR_STACK * 64];
    const u32 gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y * blockDim.y + threadIdx.y;
    if (gx >= width || gy >= height) return;
    u32* stack = &s_stack[TR_STACK * (blockDim.x * threadIdx.y + threadIdx.x)];
    const u32 index = gx * width + gy;                                             // :245 (square images)
    const RayRec ray = rays[index];
    const XformRec tr = *xf;
    u32 node = root, top = 0;
    stack[top++] = INV;
    u32 hit_prim = INV; float hit_t = FMAX, hu = 0.f, hv = 0.f;
    const F3 o = inv_transform(ray.o, tr.s, tr.q, tr.t), d = inv_transform(ray.d, tr.s, tr.q, { 0, 0, 0 });
    const F3 inv = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    while (node != INV) {
        while (node < n_internal) {
            const u32 l = nodes[node].left, r = nodes[node].right;
            float n0, f0, n1, f1;
            slab(nodes[l].aabb, o, inv, hit_t, n0, f0);
            slab(nodes[r].aabb, o, inv, hit_t, n1, f1);
            const bool hl = n0 <= f0, hr = n1 <= f1;
            if (hl || hr) {
                if (hl && hr) { node = (n0 < n1) ? l : r; if (top < (u32)TR_STACK) stack[top++] = (n0 < n1) ? r : l; }
                else node = hl ? l : r;
                continue;
            }
            node = stack[--top];
        }
        while (node >= n_internal && node != INV) {
            const u32 prim = nodes[node].left;
            const bvh_triangle t = tris[prim];
            const F3 v0 = transform({ t.v1.x, t.v1.y, t.v1.z }, tr.s, tr.q, tr.t), v1 = transform({ t.v2.x, t.v2.y, t.v2.z }, tr.s, tr.q, tr.t),
                     v2 = transform({ t.v3.x, t.v3.y, t.v3.z }, tr.s, tr.q, tr.t);
            // intersectTriangle :516-531
            const F3 p0 = sub(v0, ray.o), p1 = sub(v1, ray.o), p2 = sub(v2, ray.o), e0 = sub(v2, v0), e1 = sub(v0, v1), e2 = sub(v1, v2);
            const F3 nrm = cross3(e1, e0);
            const float u = dot3(cross3(add(p0, p2), e0), ray.d), v = dot3(cross3(add(p1, p0), e1), ray.d), w = dot3(cross3(add(p2, p1), e2), ray.d);
            const float tt = dot3(p0, nrm) * 2.0f, den = dot3(nrm, ray.d) * 2.0f;
            const float iu = u / den, iv = v / den, iw = w / den, it = tt / den;
            if (iu > 0.0f && iv > 0.0f && iw > 0.0f && it > 0.0f && it < hit_t) { hit_prim = prim; hit_t = it; hu = iu; hv = iv; }
            node = stack[--top];
        }
    }
    if (hit_prim != INV) {                                                         // :444-450
        rgba[index * 4 + 0] = to_u8(hu * 255); rgba[index * 4 + 1] = to_u8(hv * 255);
        rgba[index * 4 + 2] = to_u8((1 - hu - hv) * 255); rgba[index * 4 + 3] = 255;
    }
}

// ---- the reference's other three traversal flavours.  All four visit the near child first and accept a hit only if it is strictly
// closer, so they test the triangles along a ray in the same order and produce the same image; they differ in how they remember
// what is left to do: a per-ray stack (if-if), a stack with one postponed leaf per lane so that the lanes of a wave reach their
// triangle tests together (speculative while-while; the wave vote spans 64 lanes here), or no stack at all (restart trail: one bit
// per level says "the near subtree is done", a pop restarts from the root — which the reference hard-codes as node 0).
// counter (optional): triangle tests per ray, as the reference's rayCounter.
struct TraceRay { F3 o, d, inv; };

__device__ __forceinline__ bool tri_test(const bvh_triangle& t, const RayRec& ray, const XformRec& tr, float& hit_t, float& hu, float& hv) {
    const F3 v0 = transform({ t.v1.x, t.v1.y, t.v1.z }, tr.s, tr.q, tr.t), v1 = transform({ t.v2.x, t.v2.y, t.v2.z }, tr.s, tr.q, tr.t),
             v2 = transform({ t.v3.x, t.v3.y, t.v3.z }, tr.s, tr.q, tr.t);
    const F3 p0 = sub(v0, ray.o), p1 = sub(v1, ray.o), p2 = sub(v2, ray.o), e0 = sub(v2, v0), e1 = sub(v0, v1), e2 = sub(v1, v2);
    const F3 nrm = cross3(e1, e0);
    const float u = dot3(cross3(add(p0, p2), e0), ray.d), v = dot3(cross3(add(p1, p0), e1), ray.d), w = dot3(cross3(add(p2, p1), e2), ray.d);
    const float tt = dot3(p0, nrm) * 2.0f, den = dot3(nrm, ray.d) * 2.0f;
    const float iu = u / den, iv = v / den, iw = w / den, it = tt / den;
    if (iu > 0.0f && iv > 0.0f && iw > 0.0f && it > 0.0f && it < hit_t) { hit_t = it; hu = iu; hv = iv; return true; }
    return false;
}

template <int KIND>   // 1 restart trail, 2 if-if, 3 speculative while-while
__global__ __launch_bounds__(64) void k_trace(const RayRec* __restrict__ rays, const bvh_triangle* __restrict__ tris,
                                              const bvh2_node* __restrict__ nodes, const XformRec* __restrict__ xf,
                                              unsigned char* __restrict__ rgba, u32* __restrict__ counter, u32 root, u32 width, u32 height, u32 n_internal) {
    __shared__ u32 s_stack[KIND == 1 ? 1 : TR_STACK * 64];
    const u32 gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y * blockDim.y + threadIdx.y;
    if (gx >= width || gy >= height) return;
    u32* stack = &s_stack[KIND == 1 ? 0 : TR_STACK * (blockDim.x * threadIdx.y + threadIdx.x)];
    const u32 index = gx * width + gy;
    const RayRec ray = rays[index];
    const XformRec tr = *xf;
    u32 hit_prim = INV, tests = 0; float hit_t = FMAX, hu = 0.f, hv = 0.f;
    const F3 o = inv_transform(ray.o, tr.s, tr.q, tr.t), d = inv_transform(ray.d, tr.s, tr.q, { 0, 0, 0 });
    const F3 inv = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    // lf: the while-while / if-if / speculative rule "left first iff its entry is strictly nearer" (:310 etc.); rf: the restart trail's
    // "right first iff its entry is strictly nearer" (:118-122) — they differ when both boxes are entered at the same distance
    auto children = [&](u32 node, u32& l, u32& r, bool& hl, bool& hr, bool& lf, bool& rf) {
        l = nodes[node].left; r = nodes[node].right;
        float n0, f0, n1, f1;
        slab(nodes[l].aabb, o, inv, hit_t, n0, f0);
        slab(nodes[r].aabb, o, inv, hit_t, n1, f1);
        hl = n0 <= f0; hr = n1 <= f1; lf = n0 < n1; rf = n0 > n1;
    };
    auto leaf = [&](u32 node) { const u32 prim = nodes[node].left; ++tests; if (tri_test(tris[prim], ray, tr, hit_t, hu, hv)) hit_prim = prim; };

    if (KIND == 1) {
        // restart trail (:27-146).  `level` is a one-hot mask of the current depth (MSB = root), `trail` has a bit set at every depth
        // whose near side is finished, `pop_level` is the depth a restart is heading back to.
        constexpr u64 TOP = 0x8000000000000000ull;
        u64 trail = TOP, level = TOP, pop_level = 0;
        u32 node = root;
        bool done = false;
        auto pop = [&]() -> bool {                                   // :32-47
            trail &= (u64)(-(long long)level); trail += level;
            const u64 t = trail >> 1;
            level = ((t - 1) ^ t) + 1;
            if (!(trail & TOP)) return true;                        // the trail overflowed past the root: traversal complete
            pop_level = level; node = 0; level = TOP;               // restart from the root (node 0 in the reference, :44)
            return false;
        };
        while (!done) {
            if (node >= n_internal) { leaf(node); done = pop(); }
            else {
                u32 l, r; bool hl, hr, lf, rf; children(node, l, r, hl, hr, lf, rf);
                if (hl && hr) {
                    const u32 near = rf ? r : l, far = rf ? l : r;
                    level >>= 1;
                    node = (trail & level) ? far : near;
                } else if (hl || hr) {
                    level >>= 1;
                    if (level != pop_level) { trail |= level; node = hr ? r : l; }
                    else done = pop();
                } else done = pop();
            }
        }
    } else if (KIND == 2) {
        u32 node = root, top = 0;
        stack[top++] = INV;
        while (node != INV) {                                        // :179-226
            if (node >= n_internal) leaf(node);
            else {
                u32 l, r; bool hl, hr, lf, rf; children(node, l, r, hl, hr, lf, rf);
                if (hl || hr) {
                    if (hl && hr) { node = lf ? l : r; if (top < (u32)TR_STACK) stack[top++] = lf ? r : l; }
                    else node = hl ? l : r;
                    continue;
                }
            }
            node = stack[--top];
        }
    } else {
        u32 node = root, top = 0, pending = INV;
        stack[top++] = INV;
        while (node != INV) {                                        // :369-441
            bool searching = true;
            while (node < n_internal) {
                u32 l, r; bool hl, hr, lf, rf; children(node, l, r, hl, hr, lf, rf);
                if (hl || hr) {
                    if (hl && hr) { node = lf ? l : r; if (top < (u32)TR_STACK) stack[top++] = lf ? r : l; }
                    else node = hl ? l : r;
                } else node = stack[--top];
                if (node != INV && node >= n_internal && pending == INV) { searching = false; pending = node; node = stack[--top]; }
                if (!__any(searching)) break;                        // every lane of the wave holds a leaf
            }
            while (pending != INV) {
                if (pending >= n_internal) leaf(pending);
                pending = INV;
                if (node != INV && node >= n_internal) { pending = node; node = stack[--top]; }
            }
        }
    }
    if (counter) counter[index] = tests;
    if (hit_prim != INV) {
        rgba[index * 4 + 0] = to_u8(hu * 255); rgba[index * 4 + 1] = to_u8(hv * 255);
        rgba[index * 4 + 2] = to_u8((1 - hu - hv) * 255); rgba[index * 4 + 3] = 255;
    }
}

void launch_generate_rays(hipStream_t s, const void* d_cam, void* d_rays, uint32_t width, uint32_t height) {
    KernelScope ks(s, "k_generate_rays");
    hipLaunchKernelGGL(k_generate_rays, dim3((width + 7) / 8, (height + 7) / 8), dim3(8, 8), 0, s, (const CameraRec*)d_cam, (RayRec*)d_rays, width, height);
}
void launch_trace_while(hipStream_t s, const void* d_rays, const void* d_tris, const void* d_nodes, const void* d_xf, void* d_rgba,
                        uint32_t root, uint32_t width, uint32_t height, uint32_t n_internal) {
    KernelScope ks(s, "k_trace_while");
    hipLaunchKernelGGL(k_trace_while, dim3((width + 7) / 8, (height + 7) / 8), dim3(8, 8), 0, s, (const RayRec*)d_rays, (const bvh_triangle*)d_tris,
                       (const bvh2_node*)d_nodes, (const XformRec*)d_xf, (unsigned char*)d_rgba, root, width, height, n_internal);
}

void launch_trace_kind(hipStream_t s, int kind, const void* d_rays, const void* d_tris, const void* d_nodes, const void* d_xf, void* d_rgba,
                       uint32_t* d_counter, uint32_t root, uint32_t width, uint32_t height, uint32_t n_internal) {
    const dim3 g((width + 7) / 8, (height + 7) / 8), b(8, 8);
#define TRACE(K) hipLaunchKernelGGL(k_trace<K>, g, b, 0, s, (const RayRec*)d_rays, (const bvh_triangle*)d_tris, (const bvh2_node*)d_nodes, (const XformRec*)d_xf, \
                                    (unsigned char*)d_rgba, d_counter, root, width, height, n_internal)
    KernelScope ks(s, "k_trace");
    if (kind == 1) TRACE(1); else if (kind == 2) TRACE(2); else TRACE(3);
#undef TRACE
}

} // namespace bvh
