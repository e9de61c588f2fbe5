// kernels.hpp — host-side launch interface of the gfx950 kernels (internal; the public boundary is include/bvh_mi355x.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bvh {

// Optional per-kernel timing (bvh_ctx_set_profiling(ctx, 2)): launchers announce each kernel (or group of launches of one
// kernel) with a KernelScope; when a recorder is installed for the calling thread it records ONE hipEvent on the launch
// stream at that point.  The time between two consecutive marks is attributed to the earlier mark's name (kernel duration
// including its launch boundary) — half the events of a begin/end pair, which matters when a build is ~70 launches.
struct KernelRecorder {
    virtual void mark(hipStream_t s, const char* name) = 0;
    virtual ~KernelRecorder() {}
};
extern thread_local KernelRecorder* g_recorder;
struct KernelScope {
    KernelScope(hipStream_t stream, const char* name) { if (g_recorder) g_recorder->mark(stream, name); }
};

constexpr int EM_BLOCK = 256;
// what the build path's first kernel clears before it starts (the stand-alone sort entry point clears the same through sort_prepare / k_prepare)
struct PrepArgs { uint32_t* hist = nullptr; uint32_t hist_words = 0; uint4* status = nullptr; uint32_t status_vecs = 0; uint32_t* counters = nullptr;
                  uint32_t* extra = nullptr; uint32_t extra_words = 0;
                  // PLOC++ builds: the iteration bookkeeping of ploc_begin (look-back status words of every iteration, state words with counts[0] = n)
                  uint4* ploc_status = nullptr; uint32_t ploc_status_vecs = 0; uint64_t* ploc_tail = nullptr; uint32_t ploc_tail_words = 0;
                  uint32_t* ploc_state = nullptr; uint32_t ploc_count = 0; };
#ifdef BVH_ABLATION
constexpr int SORT_COUNTER_CLEAR = 64;      // (+ the look-back statistics of the measurement build)
#else
constexpr int SORT_COUNTER_CLEAR = 8;       // = SORT_MAX_PASSES tile tickets
#endif

// ---- stage E / M (stage_em.hip)
// reset_scene: launch the Aabb::reset of d_scene first (false when sort_prepare already did it)
void launch_extents(hipStream_t s, const void* d_tris, uint32_t n, void* d_boxes, void* d_scene, bool reset_scene = true, const PrepArgs* prep = nullptr);
void launch_extents_packed(hipStream_t s, const void* d_tris36, uint32_t n, void* d_boxes, void* d_scene, bool reset_scene = true, const PrepArgs* prep = nullptr);
void launch_extents_indexed(hipStream_t s, const void* d_vertices, const void* d_indices, uint32_t n_vertices, uint32_t n, void* d_boxes, void* d_scene, bool reset_scene = true, const PrepArgs* prep = nullptr);
void launch_morton(hipStream_t s, const void* d_boxes, uint32_t n, const void* d_scene, uint32_t* d_keys, uint32_t* d_vals,
                   uint32_t* d_hist /*may be null*/, int hist_bits, int passes, float* d_reset_next = nullptr /* Aabb::reset of another extent (the next build's) */);
// extended Morton code with a 60-bit budget in u64 keys (total_bits = 30 reproduces launch_morton's codes: the parity pin)
void launch_morton64(hipStream_t s, const void* d_boxes, uint32_t n, const void* d_scene, uint64_t* d_keys, int total_bits,
                     uint32_t* d_hist /*may be null*/, int passes, float* d_reset_next = nullptr);

// the per-scene bit plan the Morton kernels derive from a scene extent, evaluated on the device: int[10] = {axis[3], bits[3], pre[2], pre_sum, swap}
void launch_morton_plan(hipStream_t s, const void* d_scene, int* d_out, int total_bits);

// ---- stage S (sort.hip): one-sweep LSD radix sort, SORT_BITS-bit digits
constexpr int SORT_BITS = 8;
constexpr int SORT_RADIX = 1 << SORT_BITS;
constexpr int SORT_BLOCK = 256;
#ifndef BVH_SORT_IPT
#define BVH_SORT_IPT 12
#endif
constexpr int SORT_IPT = BVH_SORT_IPT;                        // keys per thread
constexpr int SORT_TILE = SORT_BLOCK * SORT_IPT;              // keys per workgroup (sizes the status rows)
constexpr int SORT_IPT_WIDE = 20;                            // keys per thread for large inputs
#ifndef BVH_SORT_WIDE_MIN_N
#define BVH_SORT_WIDE_MIN_N 1000000
#endif
constexpr uint32_t SORT_WIDE_MIN_N = BVH_SORT_WIDE_MIN_N;
constexpr int SORT_MAX_PASSES = 8;                          // 8 digits: 64-bit keys
static_assert(SORT_COUNTER_CLEAR >= SORT_MAX_PASSES, "the build path's first kernel clears the tile tickets of every pass");
// The digit histograms exist in SORT_HIST_COPIES copies (copy c at hist + c * SORT_HIST_STRIDE): a producer workgroup flushes its counts
// into copy blockIdx % copies, a sort tile adds the copies up.  One address takes ~90 atomics/us on this chip: 2048 workgroups flushing
// into ONE copy spent 23 us queueing on every bin (k_morton 88 us at 10 M, of which the stream is 45); with 16 copies it is 1.5 us.
constexpr int SORT_HIST_COPIES = 16;
constexpr int SORT_HIST_STRIDE = SORT_MAX_PASSES * (1 << SORT_BITS);
// The build's u32 sort runs on key bits [0, 30) (8/8/8/6).  The extended Morton code is the reference's unsigned wrap-around arithmetic
// (src/CommonBlocksKernel.h:252-356) and does NOT stay below 2^30 on every scene (a planar scene with an axis ratio >= 2^32: ADVICE r04) — the reference sorts
// all 32 bits, so such keys must sort by them.  k_morton raises this word (row 4 of histogram copy 0: unused by a four-pass sort, cleared with the histograms)
// when a code has bit 30 or 31 set; the narrow last pass then returns at once and the 8-bit instantiation enqueued behind it (a no-op otherwise) does the pass.
constexpr int SORT_WIDE_FLAG_WORD = 4 * (1 << SORT_BITS);
struct SortScratch {
    void*     pairs0;        // interleaved {key,value} records of the intermediate passes, ping (8 B x n for u32 keys, 16 B x n for u64)
    void*     pairs1;        // pong
    uint32_t* hist;          // u32[SORT_HIST_COPIES * SORT_HIST_STRIDE]: per copy, per pass, per digit (zeroed by sort_prepare)
    uint32_t* status;        // u32[SORT_MAX_PASSES * tiles * SORT_RADIX] (zeroed by sort_prepare)
    uint32_t* counters;      // u32[SORT_MAX_PASSES]                 (zeroed by sort_prepare)
    int       test_knobs;    // BVH_OPT_SORT_TEST_KNOBS (8 | 32): forces the helping path; results unchanged
};
inline uint32_t sort_tiles(uint32_t n) { return (n + SORT_TILE - 1) / SORT_TILE; }
inline int sort_passes(int start_bit, int end_bit) { return (end_bit - start_bit + SORT_BITS - 1) / SORT_BITS; }
size_t sort_status_bytes(uint32_t n);
// zero hist/status/counters for `passes` digits (must precede the histogram producer) — one launch that can also reset a scene
// extent (Aabb::reset) and zero one more small word array
void sort_prepare(hipStream_t s, const SortScratch& sc, uint32_t n, int passes, float* d_scene_reset = nullptr,
                  uint32_t* d_extra = nullptr, uint32_t extra_words = 0);
// hist_ready: sc.hist already holds the per-pass digit counts (fused into the Morton kernel); else a histogram kernel runs.
// gated_narrow_top (the build's u32 path only; needs hist_ready and the flag word maintained by k_morton): see SORT_WIDE_FLAG_WORD
void sort_pairs(hipStream_t s, const SortScratch& sc, const uint32_t* keys_in, const uint32_t* vals_in, uint32_t n,
                uint32_t* keys_out, uint32_t* vals_out, int start_bit, int end_bit, bool hist_ready, bool gated_narrow_top = false);
void sort_pairs64(hipStream_t s, const SortScratch& sc, const uint64_t* keys_in, const uint32_t* vals_in, uint32_t n,
                  uint64_t* keys_out, uint32_t* vals_out, int start_bit, int end_bit, bool hist_ready);

// ---- stage B (lbvh.hip, hploc.hip, ploc.hip)
// key_bits: 32 = u32 sorted keys (30-bit Morton codes, the reference), 64 = u64 sorted keys (60-bit codes)
// d_slots: u64[n] hand-off words, all-zero before the call and left all-zero (self-cleaning).  d_queue / d_queue_count: scratch of the
// tile scheduler used for large n (uint4[lbvh_queue_capacity(n)], u32[64 * 32]); pass d_queue = nullptr to force the one-launch kernel.
size_t lbvh_queue_capacity(uint32_t n);
void launch_lbvh_single(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                        void* d_nodes, uint64_t* d_slots /*u64[n]*/, uint32_t* d_root, void* d_queue, size_t queue_capacity, uint32_t* d_queue_count,
                        bool heads_cleared = false /* d_queue_count is already zero */, int scheduler = 0 /* BVH_OPT_LBVH_SCHEDULER */);
// small inputs: k_karras + k_refit (d_parent u32[2n-1]; d_flags u32[n], all 0xFFFFFFFF before the call and left so); large inputs: the tile
// scheduler with the two-pass numbering (d_slots / d_root / d_queue / d_queue_count as for launch_lbvh_single)
void launch_lbvh_two(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                     void* d_nodes, uint32_t* d_parent /*u32[2n-1]*/, uint32_t* d_flags /*u32[n]*/, uint64_t* d_slots, uint32_t* d_root,
                     void* d_queue, size_t queue_capacity, uint32_t* d_queue_count, bool heads_cleared = false, int scheduler = 0);
// HPLOC scratch (hploc.hip).  dep must be all-zero before a build and is left all-zero by a completed build.
struct HplocScratch {
    void*     recs;          // 32-byte survivor records {id, rep, box} x n
    uint64_t* dep;           // u64[n] dependency words {count:2 | R:30 | L:30}
    uint32_t* zero_parent;   // u32[1]
    uint32_t* queue_pc;      // u32[queue_capacity]   (block-local mode: nodes ready for k_hploc_ext)
    uint64_t* queue_rng;     // u64[queue_capacity]
    uint32_t* queue_count;   // u32[64 * 32 + 32]     (one padded head per sub-queue + the overlapped schedule's "tiles done" word)
    size_t    queue_capacity;
    const void* leaf_tris = nullptr;   // build path, 64-byte triangles only: the emitters stage a leaf's box from its triangle (one aligned 64-byte line per
                                       // leaf) instead of from the 24-byte box array (a box straddles two 64-byte lines one time in four); nullptr: boxes
};
size_t hploc_queue_capacity(uint32_t n);
uint32_t hploc_block_tile();
uint32_t hploc_head_words();     // words of queue_count every build starts from zero (the build path's first kernel clears them)
// Overlapped schedule of the tile scheduler (hploc.hip "k_hploc_live"): the external climb runs on `side` beside the tile kernel.  queue_pc / queue_rng must be
// all-zero before the build and are left all-zero by it (the classic schedule leaves its items behind: api.hip keeps track).
struct HplocLive { hipStream_t side; hipEvent_t fork, join; };
void launch_hploc(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                  void* d_nodes, void* d_leaves, const HplocScratch& sc);
void launch_hploc_block(hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const uint32_t* d_svals, uint32_t n,
                        void* d_nodes, void* d_leaves, const HplocScratch& sc, bool heads_cleared = false /* sc.queue_count is already zero */,
                        const HplocLive* live = nullptr /* non-null with a side stream: the overlapped schedule */);
struct PlocScratch {
    void*     list0;         // 32-byte cluster entries {id, box} x n (ping)
    void*     list1;         // pong
    uint32_t* ids1;          // u32[n]
    uint64_t* status;        // u64[PLOC_MAX_ITERS * chunks]
    uint32_t* state;         // u32[PLOC_STATE_WORDS]
    bool      static_ids = true;   // grids of <= 256 workgroups take chunk = workgroup id (no ticket); false: tickets always (BVH_OPT_PLOC_SCHEDULER 1: hosts that share the device)
};
constexpr int PLOC_CHUNK = 1024;
constexpr int PLOC_MAX_ITERS = 96;
constexpr int PLOC_STATE_WORDS = 2 * PLOC_MAX_ITERS + 4;     // counts[MAX+1] | tickets[MAX] | iterations done | (spare)
inline uint32_t ploc_chunks(uint32_t n) { return (n + PLOC_CHUNK - 1) / PLOC_CHUNK; }
void ploc_begin(hipStream_t s, const PlocScratch& sc, uint32_t n);
void ploc_begin_prep(const PlocScratch& sc, uint32_t n, PrepArgs& prep);   // the same clearing as fields of the build path's first kernel (no launch)
void ploc_reset(hipStream_t s, const PlocScratch& sc, uint32_t n, uint32_t count);
// fresh: iteration `first` is the build's very first one — it reads d_svals / d_boxes and writes d_leaves (SetupClusters fused)
void ploc_enqueue(hipStream_t s, const PlocScratch& sc, uint32_t n, void* d_nodes, void* d_leaves, const void* d_boxes, const uint32_t* d_svals,
                  int first, int count, int parity, bool fresh);

// ---- BVH2 -> BVH4 collapse (collapse.hip)
constexpr int COLLAPSE_MAX_BATCH = 64;                       // levels per batch of launches (one counter word per level)
constexpr int COLLAPSE_STATE_WORDS = COLLAPSE_MAX_BATCH;
void collapse_begin(hipStream_t s, uint2* d_taskq, uint32_t* d_state, uint32_t root, bool with_root);
void collapse_enqueue(hipStream_t s, const void* d_nodes, const void* d_leaves, void* d_wide, void* d_prims, uint2* d_taskq,
                      uint32_t* d_state, uint32_t base_begin, uint32_t base_len, int count, uint32_t n, int layout,
                      const uint32_t* expect = nullptr /* count words: the levels' task counts of the previous same-size collapse (grid sizing only) */);

// ---- consumer side (trace.hip)
void launch_generate_rays(hipStream_t s, const void* d_cam, void* d_rays, uint32_t width, uint32_t height);
void launch_trace_while(hipStream_t s, const void* d_rays, const void* d_tris, const void* d_nodes, const void* d_xf, void* d_rgba,
                        uint32_t root, uint32_t width, uint32_t height, uint32_t n_internal);

// kind: 1 restart trail, 2 if-if, 3 speculative while-while; d_counter (optional): triangle tests per ray
void launch_trace_kind(hipStream_t s, int kind, const void* d_rays, const void* d_tris, const void* d_nodes, const void* d_xf, void* d_rgba,
                       uint32_t* d_counter, uint32_t root, uint32_t width, uint32_t height, uint32_t n_internal);

// ---- helpers (misc.hip)
void launch_to_lbvh_layout(hipStream_t s, const void* d_nodes, const void* d_leaves, uint32_t n, void* d_out);
void launch_copy_bytes(hipStream_t s, void* d_dst, const void* d_src, size_t bytes);
// na words at d_a, then nb words at d_b (either may be empty), written as pairs {v, ~v} into 2 (na + nb) device-accessible pinned host words
void launch_readback(hipStream_t s, const uint32_t* d_a, uint32_t na, const uint32_t* d_b, uint32_t nb, uint32_t* pinned_pairs);      // device-to-device copy as a kernel (no runtime copy path)
void launch_sah_cost(hipStream_t s, const void* d_nodes, const void* d_leaves, uint32_t root, uint32_t n, int layout, double* d_out /*[1], zeroed inside*/);

void launch_bvh4_cost(hipStream_t s, const void* d_wide, uint32_t n_wide, const void* d_prims, const void* d_prim_boxes, uint32_t n, double* d_out /*[1], zeroed inside*/);
void launch_checksum(hipStream_t s, const void* d_nodes, uint32_t n_nodes, const void* d_leaves /*may be null*/, uint32_t n_leaves, uint32_t root, uint64_t* d_out /*[1], zeroed inside*/);

// one kernel of each translation unit of the build path is touched (hipFuncGetAttributes): the runtime loads that unit's code object now instead of at its first launch
void warm_stage_em(); void warm_sort(); void warm_lbvh(); void warm_hploc(); void warm_ploc(); void warm_misc(); void warm_collapse();

} // namespace bvh
