/* bvh_mi355x.h — C ABI of the MI355X-native BVH build path (extent -> Morton -> radix sort -> hierarchy emit).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each entry point names the
 * reference interface it replaces (paths relative to the reference repo root).  The C++ classes in
 * include/bvh/builders.hpp (same names/signatures as the reference's builders) are thin wrappers over these.
 *
 * Conventions
 *   - every function returns 0 on success, a negative value on failure (-(int)hipError_t, or BVH_E_*);
 *     nothing throws, nothing prints (the reference's CHECK_ORO logs and continues, src/Error.cpp:9-17).
 *   - a bvh_ctx is bound to one device and one HIP stream; calls on one ctx are serialised on its stream;
 *     different ctxs are independent (one ctx per GPU / per host thread for the batched builder).
 *   - "d_" pointers are device pointers on the ctx's device.  Layouts are those of include/bvh/types.h
 *     (= src/Common.h:310-441,574-578 of the reference).
 *   - there is NO CPU fallback: if the HIP runtime or a gfx950 device is missing, bvh_ctx_create fails.
 */
#ifndef BVH_MI355X_H
#define BVH_MI355X_H

#include <stdint.h>
#include "bvh/types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define BVH_E_INVALID_ARG   (-10001)
#define BVH_E_TOO_LARGE     (-10002)   /* n >= 2^30 (status words of the one-sweep sort carry 30-bit counts) */
#define BVH_E_NOT_BUILT     (-10003)
#define BVH_E_INTERNAL      (-10004)   /* a device-side consistency check failed (e.g. HPLOC node count != n-1) */

typedef struct bvh_ctx bvh_ctx;

/* Replaces Context::Context() (src/Context.cpp:7-15: device 0 hard coded) — here any device, own stream.  The first context of a process on a device also
 * loads the build path's code objects (~2.6 ms, once), so that no build pays for it: the reference compiles its kernels at this point (hiprtc, seconds). */
int  bvh_ctx_create(int device, bvh_ctx** out);
/* Same, but work is enqueued on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int  bvh_ctx_create_on_stream(int device, void* hip_stream, bvh_ctx** out);
void bvh_ctx_destroy(bvh_ctx* ctx);
/* Pre-size the ctx's device arena for builds of up to n primitives (otherwise grown lazily, outside timed regions). */
int  bvh_ctx_reserve(bvh_ctx* ctx, uint32_t n);
int  bvh_ctx_device(const bvh_ctx* ctx);
void* bvh_ctx_stream(const bvh_ctx* ctx);

/* Per-context options.  The library never reads environment variables: which scheduler a builder uses is decided by the input size unless
 * the host says otherwise here (tests and A/B measurements do).  Unknown option or value: BVH_E_INVALID_ARG, nothing changes. */
typedef enum {
    BVH_OPT_HPLOC_SCHEDULER = 0,   /* 0 auto (by n; default), 1 one asynchronous launch (k_hploc), 2 tile kernel, then the external climb (k_hploc_block / k_hploc_ext),
                                      3 tile kernel with the external climb running beside it on the context's side stream (k_hploc_block / k_hploc_live).  Same trees. */
    BVH_OPT_LBVH_SCHEDULER  = 1,   /* 0 auto (by n; default), 1 one-launch kernels (k_lbvh_single / k_karras + k_refit), 2 tile scheduler (k_lbvh_block / k_lbvh_ext) */
    BVH_OPT_SORT_TEST_KNOBS = 2,   /* bit mask, default 0; results are identical for every value.  8: tiles are handed out in reverse order; 32: threads help at
                                      the first empty poll (both force the one-sweep sort's helping path, which in-order dispatch never takes) */
    BVH_OPT_PLOC_SCHEDULER  = 3    /* 0 auto (default): one launch per iteration (device-side loop, single-workgroup tail); iterations of at most 256 chunks take chunk =
                                      workgroup id, which assumes that the hardware dispatches a grid's workgroups in id order (it does).  1: chunk tickets in every
                                      iteration — no such assumption; for hosts that share the device between contexts (the batched builder sets it on its lanes).
                                      Same trees.  (2 was a cooperative launch, removed in round 5: still accepted, means 0.) */
} bvh_option;
int  bvh_ctx_set_option(bvh_ctx* ctx, bvh_option option, int64_t value);
int  bvh_ctx_get_option(const bvh_ctx* ctx, bvh_option option, int64_t* value_out);

/* Builder selection = the reference's compile-time switch in src/main.cpp:18-22. */
typedef enum {
    BVH_LBVH_TWOPASS    = 0,   /* TwoPassLbvh::build    src/TwoPassLbvh.cpp:17-197    */
    BVH_LBVH_SINGLEPASS = 1,   /* SinglePassLbvh::build src/SinglePassLbvh.cpp:17-188 */
    BVH_PLOCPP          = 2,   /* PLOCNew::build        src/PLOC++Bvh.cpp:16-196      */
    BVH_HPLOC           = 3    /* HPLOC::build          src/Hploc.cpp:16-165          */
} bvh_algo;

/* Per-stage times, same tokens as the reference's Timer (src/Common.h:418-427, src/Timer.h:31-73).
 * ms_total = extents + morton + sort + build  (the reference's "Total Time", src/TwoPassLbvh.cpp:308-309).
 * Filled only when profiling is enabled on the ctx (bvh_ctx_set_profiling); otherwise all zero. */
typedef struct {
    float    ms_extents, ms_morton, ms_sort, ms_build, ms_collapse, ms_total;
    uint32_t ploc_iterations;          /* PLOC++: NN/merge rounds executed on device */
    uint32_t sampled;                  /* 1: the ms_* fields of this build were recorded; 0: they are zero — profiling is off, or
                                          bvh_ctx_set_kernel_sampling(ctx, k) made this one of the k-1 un-instrumented builds (hosts that
                                          average ms_* over builds must skip those) */
    uint64_t bytes_algorithmic;        /* n x the NOMINAL per-primitive figure of SURVEY.md 8(d) (two-pass 384, single-pass 420, PLOC++ 438, HPLOC 386): mesh
                                          independent.  The exact figure of a PLOC-family build depends on the mesh's cluster loads / stores, which only the
                                          oracle counts (profiles/algorithmic_bytes.json, e.g. 387.98 for the 10 M uniform HPLOC build); bench.py prices its
                                          roofline with the exact figure when that file covers the workload and says which one it used */
} bvh_timings;
/* level 0: no events (bvh_build fully asynchronous where it can be); 1: one event per stage (the reference's Timer tokens);
 * 2: additionally one event pair around every kernel launch, summed by bvh_ctx_kernel_times. */
int  bvh_ctx_set_profiling(bvh_ctx* ctx, int level);
/* Per-kernel HIP-event times accumulated since the last bvh_ctx_set_profiling(ctx, 2): returns the number of distinct
 * kernels k; names_out receives k '\n'-separated names, ms_out[k] summed milliseconds, count_out[k] launch counts. */
/* With profiling level 2, record events for ONE kernel only (its name as reported by bvh_ctx_kernel_times; NULL or "" = all kernels):
 * two events per build instead of one per launch, so that measuring the dominant kernel inside a timed region does not stretch it. */
int  bvh_ctx_set_kernel_filter(bvh_ctx* ctx, const char* kernel_name);
/* Record events (stage events of level 1 AND per-kernel events of level 2) in every `every`-th build only (default 1 = every build).  An event
 * between two launches costs a few microseconds of launch gap and the stage times need a host wait at the end of the build; sampling keeps
 * the measurement inside a timed region without stretching it.  The other builds return bvh_timings with sampled = 0 and all ms_* zero. */
int  bvh_ctx_set_kernel_sampling(bvh_ctx* ctx, uint32_t every);
int  bvh_ctx_kernel_times(bvh_ctx* ctx, char* names_out, uint32_t names_cap, float* ms_out, uint32_t* count_out, uint32_t max_kernels);

/* Result of a build.  All pointers are device pointers owned by the ctx; they stay valid until the next
 * bvh_build on the same ctx or bvh_ctx_destroy.  Mirrors the public members of the reference builders
 * (src/TwoPassLbvh.h:19-31, src/PLOC++Bvh.h:19-32):
 *   LBVH layouts : d_nodes = Bvh2Node[2n-1], internal [0,n-1), leaf i at n-1+i {left=primIdx,right=INVALID}; d_leaves = NULL
 *   PLOC layouts : d_nodes = Bvh2Node[n-1], d_leaves = PrimRef[n] in Morton order, child >= n-1 -> leaves[child-(n-1)], root 0 */
typedef struct {
    void*    d_nodes;
    void*    d_leaves;
    void*    d_prim_aabbs;        /* Aabb[n] by original primitive index (d_triangleAabb) */
    void*    d_scene_extent;      /* Aabb[1] (d_sceneExtents) */
    void*    d_sorted_keys;       /* u32[n] (d_sortedMortonCodeKeys)   */
    void*    d_sorted_vals;       /* u32[n] (d_sortedMortonCodeValues) */
    uint32_t root;                /* BVH2 root index (m_rootNodeIdx; single-pass LBVH: data dependent) */
    uint32_t n_internal;          /* n - 1 (m_nInternalNodes) */
    uint32_t n_leaves;            /* n */
    uint32_t layout;              /* 0 = LBVH layout, 1 = PLOC layout */
    uint32_t key_bits;            /* 32: d_sorted_keys is u32[n] (30-bit codes, the reference); 64: u64[n] (60-bit codes, bvh_build_ex) */
    uint32_t reserved;
    const void* d_tris;           /* the triangles the build read, on the device (d_triangleBuff): the ctx's H2D copy of a host input, or the
                                     caller's own device buffer; format as given to the build (Triangle[n] for bvh_build) */
    void*    d_morton_keys;       /* unsorted Morton codes by primitive index, u32[n] / u64[n] (d_mortonCodeKeys).  d_mortonCodeValues is
                                     not materialised: value i = i (src/CommonBlocksKernel.h:384), produced inside the first sort pass */
} bvh_result;

/* X::build(Context&, std::vector<Triangle>&).  tris: Triangle[n], 64-byte stride, host (tris_on_device = 0: copied H2D
 * into the ctx arena, untimed, as src/TwoPassLbvh.cpp:19-20 does) or device (tris_on_device = 1: used in place). n >= 2. */
int  bvh_build(bvh_ctx* ctx, bvh_algo algo, const void* tris, uint32_t n, int tris_on_device,
               bvh_result* out, bvh_timings* timings /* may be NULL */);

/* ---- extended build (no counterpart in the reference) -----------------------------------
 * Input formats that do not pay for the reference's 64-byte padded Triangle records (stage E reads 36 or ~18 bytes per
 * triangle instead of 64), and 60-bit Morton codes in u64 keys (the same extended-code arithmetic as
 * computeExtendedMortonCode with a 60-bit budget; 8 one-sweep passes; the emitters compare 96-bit {key, position} words).
 * All pointers are DEVICE pointers.  Trees built from the same triangles are identical across input formats. */
typedef enum {
    BVH_TRI_PADDED64 = 0,   /* Triangle[n], 64-byte stride (src/Common.h:429-434) — the reference layout */
    BVH_TRI_PACKED36 = 1,   /* float[9n]: v1 v2 v3 per triangle, 36-byte stride; d_tris 16-byte aligned */
    BVH_TRI_INDEXED  = 2    /* float[3*n_vertices] + uint32[3n] */
} bvh_tri_format;
typedef struct {
    uint32_t    tri_format;      /* bvh_tri_format */
    uint32_t    morton_bits;     /* 30 (reference) or 60 */
    const void* d_tris;          /* PADDED64 / PACKED36 */
    const void* d_vertices;      /* INDEXED */
    const void* d_indices;       /* INDEXED */
    uint32_t    n_vertices;      /* INDEXED (indices >= n_vertices are read as vertex 0, never out of bounds) */
    uint32_t    reserved;
} bvh_build_input;
int  bvh_build_ex(bvh_ctx* ctx, bvh_algo algo, const bvh_build_input* in, uint32_t n, bvh_result* out, bvh_timings* timings /* may be NULL */);
/* stage E on any input format */
int  bvh_stage_extents_ex(bvh_ctx* ctx, const bvh_build_input* in, uint32_t n, void* d_prim_aabbs, void* d_scene_extent);
/* stage M with a total_bits budget (<= 60) into u64 keys; total_bits = 30 reproduces bvh_stage_morton's codes */
int  bvh_stage_morton64(bvh_ctx* ctx, const void* d_prim_aabbs, uint32_t n, const void* d_scene_extent, uint64_t* d_keys, int total_bits);
/* The per-scene bit plan of stage M exactly as the device evaluates it (src/CommonBlocksKernel.h:162-275: axis order by extent, pre-bits from (int)log2f of the
 * extent ratios, the bit budget): plan_out = {axis[3], bits[3], pre[2], pre_sum, swap}.  total_bits 30 = bvh_stage_morton, <= 60 = bvh_stage_morton64.  A host
 * that must reproduce the codes bit for bit (an oracle, a CPU fallback of its own) takes the plan from here instead of evaluating log2f with another math
 * library: OCML and a host libm may truncate differently when a ratio sits within an ulp of a power of two.  Blocking (one small read-back). */
int  bvh_stage_morton_plan(bvh_ctx* ctx, const void* d_scene_extent, int total_bits, int32_t plan_out[10]);
/* stage S on u64 keys, key bits [start_bit, end_bit) with end_bit <= 64 */
int  bvh_sort_pairs64(bvh_ctx* ctx, const uint64_t* d_keys_in, const uint32_t* d_vals_in, uint32_t n,
                      uint64_t* d_keys_out, uint32_t* d_vals_out, int start_bit, int end_bit);

/* ---- stage-level entry points (one per reference kernel / library call on the path) -------------------------- */

/* CalculateSceneExtents (src/CommonBlocksKernel.h:92-114): Triangle[n] -> Aabb[n] + scene Aabb.
 * d_scene_extent is reset to {+FltMax,-FltMax} first (src/PLOC++Bvh.cpp:23-25). */
int  bvh_stage_extents(bvh_ctx* ctx, const void* d_tris, uint32_t n, void* d_prim_aabbs, void* d_scene_extent);
/* CalculateMortonCodes (src/CommonBlocksKernel.h:374-385): 30-bit extended Morton keys, values = 0..n-1. d_vals may be NULL. */
int  bvh_stage_morton(bvh_ctx* ctx, const void* d_prim_aabbs, uint32_t n, const void* d_scene_extent,
                      uint32_t* d_keys, uint32_t* d_vals);
/* Oro::RadixSort::sort(KeyValueSoA src, KeyValueSoA dst, n, startBit, endBit, stream)
 * (call sites src/TwoPassLbvh.cpp:71-89 ...): stable ascending LSD radix sort on key bits [start_bit, end_bit).
 * d_vals_in == NULL sorts (key, index) pairs.  src is not modified. */
int  bvh_sort_pairs(bvh_ctx* ctx, const uint32_t* d_keys_in, const uint32_t* d_vals_in, uint32_t n,
                    uint32_t* d_keys_out, uint32_t* d_vals_out, int start_bit, int end_bit);
/* InitBvhNodes + BvhBuildAndFit (src/SinglePassLbvhKernel.h:27-126) -> Bvh2Node[2n-1], *root_out = root index */
int  bvh_emit_lbvh_single(bvh_ctx* ctx, const void* d_prim_aabbs, const uint32_t* d_sorted_keys,
                          const uint32_t* d_sorted_vals, uint32_t n, void* d_nodes, uint32_t* root_out);
/* InitBvhNodesPrimRef + BvhBuild + FitBvhNodes (src/TwoPassLbvhKernel.h:164-235) -> Bvh2Node[2n-1], root 0 */
int  bvh_emit_lbvh_two(bvh_ctx* ctx, const void* d_prim_aabbs, const uint32_t* d_sorted_keys,
                       const uint32_t* d_sorted_vals, uint32_t n, void* d_nodes);
/* SetupClusters + Ploc/SinglePassPloc + host loop (src/Ploc++Kernel.h:39-362, src/PLOC++Bvh.cpp:132-152) */
int  bvh_emit_ploc(bvh_ctx* ctx, const void* d_prim_aabbs, const uint32_t* d_sorted_vals, uint32_t n,
                   void* d_nodes, void* d_leaves, uint32_t* iterations_out);
/* SetupClusters + HPloc (src/HplocKernel.h:39-315) */
int  bvh_emit_hploc(bvh_ctx* ctx, const void* d_prim_aabbs, const uint32_t* d_sorted_keys,
                    const uint32_t* d_sorted_vals, uint32_t n, void* d_nodes, void* d_leaves);

/* ---- consumers' helpers --------------------------------------------------------------------------------------- */
/* PLOC layout -> LBVH layout (Bvh2Node[2n-1]) so that the reference's traversal kernels (src/TraversalKernel.h) can
 * consume PLOC/HPLOC trees; the adapter the reference never wrote. */
int  bvh_to_lbvh_layout(bvh_ctx* ctx, const bvh_result* in, void* d_nodes_2n_minus_1);
/* CollapseToWide4Bvh (src/TwoPassLbvhKernel.h:237-336, src/Ploc++Kernel.h:364-465) + host set-up (src/TwoPassLbvh.cpp:154-183):
 * BVH2 -> BVH4.  d_bvh4: Bvh4Node[n] (128 B each, src/Common.h:560-566), d_primnodes: PrimNode[n] (src/Common.h:568-572);
 * wide root = node 0; *n_wide_out = number of wide nodes.  Blocking (reads the level bounds back). */
int  bvh_collapse4(bvh_ctx* ctx, const bvh_result* in, void* d_bvh4, void* d_primnodes, uint32_t* n_wide_out);
/* duration of the last bvh_collapse4 on this ctx (the reference's CollapseBvhTime token); 0 unless profiling was on */
int  bvh_ctx_last_collapse_ms(const bvh_ctx* ctx, float* ms_out);
/* ---- consumer side used by the image check ---- */
/* GenerateRays (src/CommonBlocksKernel.h:432-463).  h_camera: 64-byte Camera record (src/Common.h:550-558) on the host;
 * d_rays: Ray[width*height] (32 bytes each, src/Common.h:533-539), ray of pixel (gx,gy) at index gx*height+gy. */
int  bvh_generate_rays(bvh_ctx* ctx, const void* h_camera, void* d_rays, uint32_t width, uint32_t height);
/* BvhTraversalWhile (src/TraversalKernel.h:238-335): while-while closest-hit traversal of an LBVH-layout Bvh2Node[2n-1] array
 * (use bvh_to_lbvh_layout for PLOC/HPLOC results).  h_transform: 64-byte Transformation (src/Common.h:541-548) on the host.
 * d_rgba: width*height*4 bytes, cleared, then u8 (u*255, v*255, (1-u-v)*255, 255) per hit pixel at index gx*width+gy.
 * Square images only (the reference indexes rays with height and pixels with width). */
int  bvh_trace_while(bvh_ctx* ctx, const void* d_rays, const void* d_tris, const void* d_nodes_lbvh, uint32_t root, uint32_t n_internal,
                     const void* h_transform, void* d_rgba, uint32_t width, uint32_t height);
/* The reference's four traversal kernels behind one entry point: BvhTraversalRestartTrail (src/TraversalKernel.h:49-146; stackless, a
 * restart re-enters at node 0, so root must be 0), BvhTraversalifif (:148-236), BvhTraversalWhile (:238-335), BvhTraversalSpeculativeWhile
 * (:337-451; the wave vote spans 64 lanes here).  Same arguments as bvh_trace_while; d_ray_counter (optional, u32[width*height]) receives
 * the triangle tests per ray (the reference's rayCounter; not counted by the while-while kernel: zeroed).  All four produce the same image. */
typedef enum { BVH_TRACE_WHILE_WHILE = 0, BVH_TRACE_RESTART_TRAIL = 1, BVH_TRACE_IF_IF = 2, BVH_TRACE_SPECULATIVE_WHILE = 3 } bvh_trace_kind;
int  bvh_trace(bvh_ctx* ctx, bvh_trace_kind kind, const void* d_rays, const void* d_tris, const void* d_nodes_lbvh, uint32_t root, uint32_t n_internal,
               const void* h_transform, void* d_rgba, uint32_t* d_ray_counter, uint32_t width, uint32_t height);
/* BVH2 SAH cost with the formula of Utility::calculateLbvhCost (src/Utility.cpp:317-349), device reduction, f64. */
int  bvh_sah_cost(bvh_ctx* ctx, const bvh_result* in, double* cost_out);
/* BVH4 cost with the formula of Utility::calculatebvh4Cost (src/Utility.cpp:351-396) — the value the reference's builders store in
 * m_cost after the collapse (src/TwoPassLbvh.cpp:196, src/SinglePassLbvh.cpp:186, src/PLOC++Bvh.cpp:195, src/Hploc.cpp:164).
 * d_bvh4 / d_primnodes / n_wide: outputs of bvh_collapse4; d_prim_aabbs: Aabb[n] by primitive index (bvh_result.d_prim_aabbs).
 * Device reduction, f32 terms, f64 accumulation.  Blocking. */
int  bvh_bvh4_cost(bvh_ctx* ctx, const void* d_bvh4, uint32_t n_wide, const void* d_primnodes, const void* d_prim_aabbs, uint32_t n, double* cost_out);
/* Order-independent 64-bit checksum of a result's node array, leaf array (PLOC layouts) and root index: equal checksums <=> byte-identical
 * results (up to hash collisions).  Lets hosts compare builds (batched vs single, run vs run) without reading the arrays back.  Blocking. */
int  bvh_checksum(bvh_ctx* ctx, const bvh_result* in, uint64_t* checksum_out);
/* copy a result's arrays to host (blocking), sizes per layout; any pointer may be NULL.  h_sorted_keys: u32[n] or, for
 * key_bits == 64 results, u64[n] */
int  bvh_download(bvh_ctx* ctx, const bvh_result* in, void* h_nodes, void* h_leaves, void* h_sorted_keys,
                  uint32_t* h_sorted_vals, void* h_scene_extent);

/* BatchedBvhBuilder::build (src/BatchedBuilder.h:12-31) re-purposed as the multi-GPU scene shard (BASELINE.json config 5), one
 * process: mesh m -> devs[m % n_dev], one ctx + host thread per device, then ONE RCCL all-gather of the root AABBs.
 * h_tris[m]: host Triangle[n_tris[m]].  root_aabbs_out: 6 floats per mesh (min xyz, max xyz).  build_ms_out (optional):
 * E+M+S+B milliseconds per mesh.  Blocking. */
int  bvh_batched_build(int n_dev, const int* devs, bvh_algo algo, const void* const* h_tris, const uint32_t* n_tris, int n_meshes,
                       float* root_aabbs_out, float* build_ms_out);
/* The same as a reusable object: the per-device contexts (arenas), the RCCL communicator and the gather buffers are created once and
 * kept across builds (bvh_batched_build pays for them on every call).  Report fields are optional except root_aabbs. */
typedef struct bvh_batch bvh_batch;
/* where one mesh's tree lives after bvh_batch_build (the reference's BatchedBvhBuilder keeps every mesh's nodes and leaves on the device: d_bvhNodes / d_primRefs /
 * d_rootNodes, src/BatchedBuilder.h:24-26).  Device memory owned by the batch, valid until its next build or bvh_batch_destroy; child indices are mesh-local. */
typedef struct {
    int32_t     device;        /* HIP device the arrays live on */
    uint32_t    n_leaves, n_internal, n_nodes, root, layout;   /* as bvh_result; n_nodes = Bvh2Node records at d_nodes (layout 0: 2n-1, layout 1: n-1) */
    const void* d_nodes;       /* Bvh2Node[n_nodes] */
    const void* d_leaves;      /* PrimRef[n_leaves] (layout 1) or NULL (layout 0: the leaves are nodes n-1 .. 2n-2) */
} bvh_batch_mesh;
typedef struct {
    float*    root_aabbs;      /* [6 * n_meshes] min xyz, max xyz per mesh, as all-gathered (device devs[0]'s copy) */
    float*    build_ms;        /* [n_meshes] or NULL: E+M+S+B per mesh (stage events) */
    uint64_t* checksums;       /* [n_meshes] or NULL: bvh_checksum of each mesh's tree */
    double*   sah;             /* [n_meshes] or NULL: bvh_sah_cost of each mesh's tree */
    float     allgather_us;    /* out: duration of the RCCL all-gather of the root boxes (HIP events, max over devices) */
    float     wall_ms;         /* out: host wall time of the call (H2D copies of the inputs included) */
    bvh_batch_mesh* meshes;    /* [n_meshes] or NULL: every mesh's tree is copied out of its context's arena and kept (ABI 4) */
    int32_t   lanes_per_device;/* out: contexts (streams + host threads) each device pipelined its meshes on: min(3, meshes per device) (ABI 4) */
    int32_t   reserved;
} bvh_batch_report;
int  bvh_batch_create(int n_dev, const int* devs, bvh_batch** out);
/* mesh m is built on devs[m % n_dev]; a device that holds several meshes pipelines them on up to three contexts (H2D of one mesh, build of another, checksum /
 * tree copy of a third overlap); one ncclAllGather of the root boxes at the end.  Blocking. */
int  bvh_batch_build(bvh_batch* batch, bvh_algo algo, const void* const* h_tris, const uint32_t* n_tris, int n_meshes, bvh_batch_report* report);
/* read one kept tree back (h_nodes: Bvh2Node[n_nodes], h_leaves: PrimRef[n_leaves] or NULL) */
int  bvh_batch_download(bvh_batch* batch, const bvh_batch_mesh* mesh, void* h_nodes, void* h_leaves);
void bvh_batch_destroy(bvh_batch* batch);

/* wait for everything enqueued on the ctx's stream (bvh_build is asynchronous unless it has to read something back:
 * profiling on, single-pass root index, PLOC++ iteration batches, collapse level counts; those read-backs are 4-byte
 * copies into pinned host words behind the build's launches on the in-order stream, and the call returns once the word
 * has landed — the host polls it, which notices the end of the build a few microseconds before hipStreamSynchronize does —
 * i.e. when every launch of the build has completed) */
int  bvh_ctx_synchronize(bvh_ctx* ctx);

/* plain device-memory helpers so that hosts without a HIP binding (ctypes, cgo, JNI ...) can stage buffers */
int  bvh_dev_alloc(bvh_ctx* ctx, uint64_t bytes, void** out);
int  bvh_dev_free(bvh_ctx* ctx, void* p);
int  bvh_dev_upload(bvh_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes);
int  bvh_dev_download(bvh_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes);
int  bvh_dev_copy(bvh_ctx* ctx, void* d_dst, const void* d_src, uint64_t bytes);   /* device->device, asynchronous on the ctx's stream */

const char* bvh_version(void);
/* ABI revision of this header (BVH_ABI_VERSION): bumped whenever a struct of this file changes size or an entry point changes signature, so that a host
 * compiled against an older header can refuse to run instead of handing the library a too-small bvh_result.  Revision 3: bvh_result carries d_tris and
 * d_morton_keys (88 bytes; round 2 grew it without a bump), bvh_ctx_set_option / bvh_ctx_get_option exist.  Revision 4: bvh_batch_report carries `meshes`
 * and `lanes_per_device` (56 bytes), bvh_batch_mesh / bvh_batch_download / bvh_stage_morton_plan exist. */
#define BVH_ABI_VERSION 4
uint32_t bvh_abi_version(void);
/* sizeof(bvh_result) / sizeof(bvh_timings) / sizeof(bvh_build_input) as the LIBRARY was compiled: out[0..2] */
void bvh_abi_struct_sizes(uint32_t out[3]);

#ifdef __cplusplus
}
#endif
#endif
