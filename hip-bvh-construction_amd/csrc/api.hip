// api.hip — the C ABI of include/bvh_mi355x.h: context/arena management and the host orchestration of the build
// (what X::build does in the reference: src/TwoPassLbvh.cpp:17-197, src/SinglePassLbvh.cpp:17-188,
// src/PLOC++Bvh.cpp:16-196, src/Hploc.cpp:16-165 — minus RTC compilation, per-call allocation and debug read-backs).
#include <hip/hip_runtime.h>
#include <new>
#include <cstring>
#include <cstdlib>
#include "bvh_mi355x.h"
#include <mutex>
#include "common.hpp"
#include "kernels.hpp"

#include <vector>
#include <string>
#include <map>

using namespace bvh;

namespace bvh { thread_local KernelRecorder* g_recorder = nullptr; }

struct EventRecorder : KernelRecorder {
    struct Rec { const char* name; hipEvent_t ev; };
    std::vector<Rec> recs; size_t used = 0;
    std::string only;                 // non-empty: record this kernel's marks only (and the mark that closes each of its intervals)
    bool open = false;
    void mark(hipStream_t s, const char* name) override {
        if (!only.empty()) {
            const bool match = name && only == name;
            if (!match && !open) return;          // an event costs ~5 us of launch gap: a build is 7 launches
            if (!match) name = nullptr;           // closes the interval of the watched kernel
            open = match;
        }
        if (used == recs.size()) { Rec r{name, nullptr}; if (hipEventCreate(&r.ev) != hipSuccess) return; recs.push_back(r); }
        recs[used].name = name; (void)hipEventRecord(recs[used].ev, s); ++used;
    }
    void reset() { used = 0; open = false; }
    ~EventRecorder() override { for (auto& r : recs) if (r.ev) (void)hipEventDestroy(r.ev); }
};

struct bvh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;           // stage-level events (the reference's Timer tokens)
    bool kernel_profiling = false;    // + one event pair per kernel launch
    uint32_t sample_every = 1, build_counter = 0;   // ... of every sample_every-th build only
    EventRecorder recorder;
    uint32_t cap = 0;                 // primitives the arena is sized for
    char* arena = nullptr;
    size_t arena_bytes = 0;
    // carved from the arena
    void* tris = nullptr;             // Triangle[cap] staging for host inputs (allocated lazily, separate)
    uint32_t tris_cap = 0;
    bvh_aabb* boxes = nullptr;
    float* scene = nullptr;
    u32 *keys = nullptr, *skeys = nullptr, *svals = nullptr;   // keys / skeys: 8 bytes per primitive (u32 keys use the first half)
    SortScratch sort{};
    bvh2_node* nodes = nullptr;       // 2*cap
    bvh_primref* leaves = nullptr;    // cap
    u64* slots = nullptr;             // cap           (u64 scratch: BVH4 collapse task queue)
    u32* parent = nullptr;            // 2*cap         (two-pass parent pointers / HPLOC parentIdx)
    u32* flags = nullptr;             // cap           (two-pass refit exchange words: all-INVALID, kept so by the protocol)
    HplocScratch hploc{};             //               (hploc.dep is zeroed when the arena is allocated and stays clean)
    PlocScratch ploc{};
    u32* small = nullptr;             // 64 words: [0] root, [1] hploc node counter, [8..9] f64 SAH
    size_t lbvh_queue_capacity = 0;   // uint4 entries of ploc.list0 (>= kernels.hpp lbvh_queue_capacity(cap))
    // The emitters' self-cleaning scratch (hploc.dep / LBVH slots all-zero, two-pass flags all-ones) is only clean after a build that ran to
    // completion.  Set while an emit is being enqueued, cleared when every launch of it was accepted: a build that failed in between
    // (HIP error, early return) makes the next one re-initialise the words instead of silently producing wrong trees.
    bool scratch_dirty = false;
    hipEvent_t ev[8] = {};
    int scene_slot = 0;               // which of the two scene extents the next build uses
    bool scene_ready = false;         // that extent holds Aabb::reset values (written by the previous build's Morton kernel); false: reset it explicitly
    uint32_t ploc_last_n = 0, ploc_last_iters = 0;   // size and iteration count of the last PLOC++ build (run_ploc aims its first batch of launches at it)
    u32* h_pinned = nullptr;          // pinned, device-accessible host words: the small read-backs (root index, PLOC++ state, collapse level counts) land here as {v, ~v} pairs
    u32* d_pinned = nullptr;          // the same words as the device addresses them
    float last_collapse_ms = 0.f;     // CollapseBvhTime of the last bvh_collapse4 (profiling on)
    uint32_t collapse_last_n = 0, collapse_last_levels = 0;   // levels the previous collapse of a tree of this size needed (first batch of launches)
    uint64_t collapse_last_key = 0;                           // ... of this KIND: {node layout, root index} — an LBVH and a PLOC tree of one size have different level widths
    uint32_t collapse_last_len[COLLAPSE_MAX_BATCH] = {0};     // ... and their task counts (grid sizes of the first batch's launches),
    int collapse_len_batch = 0;                               //     valid for that many levels (0: the previous collapse took several batches)
    hipStream_t side = nullptr;       // the overlapped HPLOC schedule's consumer stream (k_hploc_live), with its fork / join events
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool queue_items_stale = false;   // the classic tile schedule leaves its queue items behind; the overlapped one needs (and leaves) the slots all-zero
    int64_t options[4] = {0, 0, 0, 0}; // bvh_option values (bvh_ctx_set_option); all default 0 = decide by input size / no test knobs
};

namespace {

// HPLOC: one asynchronous launch below this size, LDS-tiled block kernel + external climb above (measured on MI355X: 1122 vs 966
// Mtris/s at 262 k, 3005 vs 3416 at 2 M, 3848 vs 5616 at 10 M; whole build, one launch / tiles, end of round 3 (tools/ab_sched_small.py):
// 400 k 0.1928 / 0.2302 ms, 600 k 0.2134 / 0.2234, 900 k 0.2603 / 0.2511)
constexpr uint32_t HPLOC_BLOCK_MIN_N = 800000;
inline bool hploc_use_block(const bvh_ctx* c, uint32_t n) {
    if (n <= 2 * hploc_block_tile()) return false;     // the root must cross tiles
    const int64_t o = c->options[BVH_OPT_HPLOC_SCHEDULER];   // 1 async / 2 tiles: the host's override (A/B measurements, tests)
    if (o == 1) return false;
    if (o == 2 || o == 3) return true;
    return n >= HPLOC_BLOCK_MIN_N;
}
// the overlapped schedule (k_hploc_live beside the tile kernel) only when the host asks for it: measured slower at every size (LEADS.md row 87)
inline bool hploc_use_live(const bvh_ctx* c, uint32_t) { return c->options[BVH_OPT_HPLOC_SCHEDULER] == 3; }
// HPLOC emit on the ctx's scratch (SetupClusters + HPloc, src/Hploc.cpp:83-121)
void emit_hploc(bvh_ctx* c, hipStream_t s, const void* d_boxes, const void* d_skeys, int key_bits, const u32* d_svals, uint32_t n, void* d_nodes, void* d_leaves,
                bool heads_cleared = false) {
    if (hploc_use_block(c, n)) {
        // the overlapped schedule's consumer stream and its two ordering events (no time stamps) exist from the first build that asks for them (a hardware queue per
        // context that never uses it would be one more for the batched builder's lanes to share)
        if (hploc_use_live(c, n) && !c->side) {
            if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) c->side = nullptr;
            else if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
                if (c->ev_fork) { (void)hipEventDestroy(c->ev_fork); c->ev_fork = nullptr; }
                (void)hipStreamDestroy(c->side); c->side = nullptr;
            }
            (void)hipGetLastError();       // (a failure falls back to the classic schedule: same trees)
        }
        const bool live = hploc_use_live(c, n) && c->side;
        const HplocLive lv{ c->side, c->ev_fork, c->ev_join };
        if (live && c->queue_items_stale) {      // (only after a switch of schedules on one context, or a build that failed half-way)
            (void)hipMemsetAsync(c->hploc.queue_pc, 0, c->hploc.queue_capacity * sizeof(u32), s);
            (void)hipMemsetAsync(c->hploc.queue_rng, 0, c->hploc.queue_capacity * sizeof(u64), s);
            c->queue_items_stale = false;
        }
        if (!live) c->queue_items_stale = true;
        launch_hploc_block(s, d_boxes, d_skeys, key_bits, d_svals, n, d_nodes, d_leaves, c->hploc, heads_cleared, live ? &lv : nullptr);
    }
    else launch_hploc(s, d_boxes, d_skeys, key_bits, d_svals, n, d_nodes, d_leaves, c->hploc);
}

inline int herr(hipError_t e) { return e == hipSuccess ? 0 : -(int)e; }
#define HIP_TRY(x) do { hipError_t _e = (x); if (_e != hipSuccess) return -(int)_e; } while (0)

// The per-build read-backs (single-pass LBVH root index, PLOC++ iteration state, the collapse's level counts) are a handful of words.  They come back as PAIRS
// {v, ~v} written by a one-workgroup kernel straight into pinned, device-accessible host words (misc.hip launch_readback); the caller arms every pair with
// {sentinel, sentinel} — an inconsistent pair — and wait_readback polls until EVERY pair reads {v, ~v}: hipStreamSynchronize notices the end of the stream ~6 us later
// than a poll of the words does (tools/probes/sync_latency.hip), 4 % of a 262 144-triangle LBVH build.  A consistent pair proves that every byte of both words is final:
// no assumption about the order or the granularity in which a copy becomes visible to the host (round 3 polled the last word of a hipMemcpyAsync for "not the
// sentinel" — ADVICE r03 —, and round 4 caught that copy landing byte by byte: see misc.hip).  Acquire loads: nothing read afterwards can be hoisted above the poll.  The
// stream is in order, so everything enqueued before the read-back kernel is complete when the pairs are.  Bounded: after ~4 M rounds it falls back to hipStreamSynchronize,
// which also surfaces an error of the stream.
static int wait_readback(hipStream_t s, const u32* pairs, u32 count) {
    auto all_landed = [&]() -> bool {
        for (u32 i = 0; i < count; ++i) {
            const u32 v = __atomic_load_n(pairs + 2 * i, __ATOMIC_ACQUIRE), w = __atomic_load_n(pairs + 2 * i + 1, __ATOMIC_ACQUIRE);
            if (w != ~v) return false;
        }
        return true;
    };
    for (u32 polls = 0; polls < (1u << 22); ++polls) {
        if (all_landed()) return 0;
#if defined(__x86_64__)
        __builtin_ia32_pause();                      // (be a polite spinner: the runtime's helper threads may share this core)
#endif
    }
    HIP_TRY(hipStreamSynchronize(s));
    return all_landed() ? 0 : BVH_E_INTERNAL;
}
static inline void arm_readback(u32* pairs, u32 count) {
    for (u32 i = 0; i < 2 * count; ++i) __atomic_store_n(pairs + i, 0xFFFFFFFFu, __ATOMIC_RELAXED);      // {S, S}: never a consistent pair
    __atomic_thread_fence(__ATOMIC_RELEASE);
}
// enqueue the read-back of na words at d_a and nb words at d_b behind everything on the stream and wait for it; the values are then pairs[2 i]
static int read_back(bvh_ctx* c, const u32* d_a, u32 na, const u32* d_b, u32 nb, u32* pairs);

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {
    char* base; size_t off = 0;
    template <typename T> T* take(size_t count) { T* p = base ? reinterpret_cast<T*>(base + off) : nullptr; off += align_up(count * sizeof(T)); return p; }
};

void carve(bvh_ctx* c, char* base, uint32_t cap, size_t* total) {
    Carver k{base};
    const size_t n = cap;
    c->boxes = k.take<bvh_aabb>(n);
    c->scene = k.take<float>(16);          // two extents of 8 floats: builds alternate, the Morton kernel of one build resets the other for the next
    c->keys = reinterpret_cast<u32*>(k.take<u64>(n)); c->skeys = reinterpret_cast<u32*>(k.take<u64>(n)); c->svals = k.take<u32>(n);
    c->sort.pairs0 = k.take<uint4>(n); c->sort.pairs1 = k.take<uint4>(n);
    c->sort.hist = k.take<u32>(SORT_HIST_COPIES * SORT_HIST_STRIDE);
    c->sort.status = k.take<u32>(sort_status_bytes(cap) / sizeof(u32));
    c->sort.counters = k.take<u32>(SORT_MAX_PASSES);
    c->nodes = k.take<bvh2_node>(2 * n);
    c->leaves = k.take<bvh_primref>(n);
    c->slots = k.take<u64>(n);
    c->parent = k.take<u32>(2 * n);
    c->flags = k.take<u32>(n);
    c->hploc.recs = k.take<bvh2_node>(n);
    c->hploc.dep = k.take<u64>(n);
    c->hploc.queue_capacity = hploc_queue_capacity(cap);
    c->hploc.queue_pc = k.take<u32>(c->hploc.queue_capacity);
    c->hploc.queue_rng = k.take<u64>(c->hploc.queue_capacity);
    c->hploc.queue_count = k.take<u32>(hploc_head_words());  // (the padded heads + the overlapped schedule's "tiles done" word)
    c->lbvh_queue_capacity = 2 * n > lbvh_queue_capacity(cap) ? 2 * n : lbvh_queue_capacity(cap);   // list0 doubles as the LBVH tile scheduler's root queue
    c->ploc.list0 = k.take<uint4>(c->lbvh_queue_capacity);
    c->ploc.list1 = k.take<uint4>(2 * n);
    c->ploc.ids1 = k.take<u32>(n);
    c->ploc.status = k.take<u64>((size_t)PLOC_MAX_ITERS * ploc_chunks(cap));
    c->ploc.state = k.take<u32>(PLOC_STATE_WORDS);
    c->small = k.take<u32>(64);            // [0] root, [1] hploc zero-parent, [8..9] f64 SAH / BVH4 cost, [10..11] u64 checksum, [16..31] camera, [32..47] transform, [48..57] Morton plan read-back
    c->hploc.zero_parent = c->small + 1;
    *total = k.off;
}

int ensure_capacity(bvh_ctx* c, uint32_t n) {
    if (n <= c->cap) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->arena) { HIP_TRY(hipFree(c->arena)); c->arena = nullptr; c->cap = 0; }
    size_t total = 0;
    carve(c, nullptr, n, &total);
    char* p = nullptr;
    HIP_TRY(hipMalloc(&p, total));
    c->arena = p; c->arena_bytes = total; c->cap = n;
    c->scene_ready = false; c->scene_slot = 0;
    carve(c, p, n, &total);
    HIP_TRY(hipMemsetAsync(c->hploc.dep, 0, (size_t)n * sizeof(u64), c->stream));   // HPLOC dependency words: clean once, builds keep them clean
    HIP_TRY(hipMemsetAsync(c->flags, 0xFF, (size_t)n * sizeof(u32), c->stream));    // two-pass LBVH exchange words: likewise
    HIP_TRY(hipMemsetAsync(c->hploc.queue_pc, 0, c->hploc.queue_capacity * sizeof(u32), c->stream));    // HPLOC queue slots (overlapped schedule: a slot is its own flag): likewise
    HIP_TRY(hipMemsetAsync(c->hploc.queue_rng, 0, c->hploc.queue_capacity * sizeof(u64), c->stream));
    c->queue_items_stale = false;
    return 0;
}

// re-initialise the self-cleaning emit scratch after a build that did not run to completion (see bvh_ctx::scratch_dirty)
int begin_emit(bvh_ctx* c) {
    if (c->scratch_dirty) {
        HIP_TRY(hipMemsetAsync(c->hploc.dep, 0, (size_t)c->cap * sizeof(u64), c->stream));
        HIP_TRY(hipMemsetAsync(c->flags, 0xFF, (size_t)c->cap * sizeof(u32), c->stream));
        c->queue_items_stale = true;       // (a consumer that was cut short may have left items behind: the next overlapped build zeroes the slots)
    }
    c->scratch_dirty = true;
    return 0;
}
int end_emit(bvh_ctx* c) {
    HIP_TRY(hipGetLastError());
    c->scratch_dirty = false;
    return 0;
}

int ensure_tris(bvh_ctx* c, uint32_t n) {
    if (n <= c->tris_cap) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->tris) { HIP_TRY(hipFree(c->tris)); c->tris = nullptr; c->tris_cap = 0; }
    HIP_TRY(hipMalloc(&c->tris, (size_t)n * sizeof(bvh_triangle)));
    c->tris_cap = n;
    return 0;
}

struct Bind { int prev = -1; bool ok = true;
    explicit Bind(int dev) { ok = hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess); }
    ~Bind() { int cur; if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) hipSetDevice(prev); } };

static int read_back(bvh_ctx* c, const u32* d_a, u32 na, const u32* d_b, u32 nb, u32* pairs) {
    arm_readback(pairs, na + nb);
    launch_readback(c->stream, d_a, na, d_b, nb, c->d_pinned + (pairs - c->h_pinned));
    HIP_TRY(hipGetLastError());
    return wait_readback(c->stream, pairs, na + nb);
}

// PLOC++ iteration driver: batches of device-side iterations, one small read-back per batch (src/PLOC++Bvh.cpp:132-152
// reads back after EVERY iteration).
int run_ploc(bvh_ctx* c, uint32_t n, void* d_nodes, void* d_leaves, const void* d_boxes, const u32* d_svals, const PlocScratch& sc, uint32_t* iterations_out) {
    u32* const host_state = c->h_pinned + 16;                // (pinned: the per-batch read-back does not go through a staging buffer)
    constexpr size_t state_bytes = PLOC_STATE_WORDS * sizeof(u32);
    u32 iters_before = 0;                                    // iterations counted before the bookkeeping was last restarted (a restart clears the device's counter)
    int first = 0, parity = 0;
    bool fresh = true;
    // iterations needed grow by ~3 per doubling of n (measured: 30 at 262 k, 45 at 10 M); the first batch aims slightly above
    // (round 2: 30 at 262 k, 34 at 2 M, 40 at 10 M on uniform meshes — two more per doubling; a launch after the end still costs ~5 us)
    int batch = 32; for (uint32_t m = n; m > 262144u; m >>= 1) batch += 2; if (batch > 80) batch = 80;
    // rebuilds of a scene of the same size (animation frames; the benchmark loop) need the same number of iterations give or take one: aim one above the
    // previous build's count instead of two to four (an iteration launched after the end costs ~5 us; one short costs a read-back and a second batch)
    if (c->ploc_last_n == n && c->ploc_last_iters > 0 && c->ploc_last_iters + 1 < PLOC_MAX_ITERS) batch = (int)c->ploc_last_iters + 1;
    // every iteration merges at least the globally closest pair, so n iterations always suffice (a collinear, zero-area scene needs
    // almost that many: every union has area 0 and only the lowest pair of a chunk is mutual); the reference loops the same way
    for (uint32_t guard = 0; guard < n / 16u + 4096u; ++guard) {
        if (first + batch > PLOC_MAX_ITERS) {
            // restart the per-iteration bookkeeping with the current count (pathologically slow convergence only)
            HIP_TRY(hipMemcpyAsync(host_state, sc.state, state_bytes, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            const u32 count = host_state[first];
            iters_before += host_state[2 * PLOC_MAX_ITERS + 1];
            parity = (parity + first) & 1;
            ploc_reset(c->stream, sc, n, count);
            first = 0;
        }
        ploc_enqueue(c->stream, sc, n, d_nodes, d_leaves, d_boxes, d_svals, first, batch, parity, fresh);
        fresh = false;
        // two words come back: the iterations done so far and the cluster count after this batch (pairs at pinned words 4..7)
        u32* const rb = c->h_pinned + 4;
        { const int wr = read_back(c, sc.state + 2 * PLOC_MAX_ITERS + 1, 1, sc.state + first + batch, 1, rb); if (wr) return wr; }
        const u32* const h_iters = rb; const u32* const h_count = rb + 2;
        const u32 count = *h_count;
        if (count <= 1) {
            c->ploc_last_n = n; c->ploc_last_iters = iters_before ? 0u : *h_iters;
            if (iterations_out) *iterations_out = iters_before + *h_iters;
            return 0;
        }
        first += batch; batch = 16;
    }
    return BVH_E_INTERNAL;
}

// stage E on any of the input formats of bvh_build_input (device pointers)
int stage_extents_valid(const bvh_build_input* in) {
    switch (in->tri_format) {
        case BVH_TRI_PADDED64: return in->d_tris ? 0 : BVH_E_INVALID_ARG;
        case BVH_TRI_PACKED36: return (in->d_tris && !((uintptr_t)in->d_tris & 15u)) ? 0 : BVH_E_INVALID_ARG;      // 16-byte loads
        case BVH_TRI_INDEXED:  return (in->d_vertices && in->d_indices && in->n_vertices) ? 0 : BVH_E_INVALID_ARG;
        default: return BVH_E_INVALID_ARG;
    }
}
int stage_extents_fmt(hipStream_t s, const bvh_build_input* in, uint32_t n, void* d_boxes, void* d_scene, bool reset_scene = true, const PrepArgs* prep = nullptr) {
    const int v = stage_extents_valid(in); if (v) return v;
    switch (in->tri_format) {
        case BVH_TRI_PADDED64: launch_extents(s, in->d_tris, n, d_boxes, d_scene, reset_scene, prep); return 0;
        case BVH_TRI_PACKED36: launch_extents_packed(s, in->d_tris, n, d_boxes, d_scene, reset_scene, prep); return 0;
        case BVH_TRI_INDEXED:  launch_extents_indexed(s, in->d_vertices, in->d_indices, in->n_vertices, n, d_boxes, d_scene, reset_scene, prep); return 0;
        default: return BVH_E_INVALID_ARG;
    }
}

// per-primitive algorithmic bytes of the whole pipeline (restated in DESIGN.md)
uint64_t algorithmic_bytes(bvh_algo a, uint32_t n) {
    static const uint64_t per_prim[4] = { 384, 420, 438, 386 };
    return per_prim[(int)a] * (uint64_t)n;
}

} // namespace

extern "C" {

const char* bvh_version(void) { return "bvh_mi355x 0.4 (gfx950)"; }
uint32_t bvh_abi_version(void) { return BVH_ABI_VERSION; }
void bvh_abi_struct_sizes(uint32_t out[3]) { if (out) { out[0] = (uint32_t)sizeof(bvh_result); out[1] = (uint32_t)sizeof(bvh_timings); out[2] = (uint32_t)sizeof(bvh_build_input); } }

int bvh_ctx_set_option(bvh_ctx* c, bvh_option option, int64_t value) {
    if (!c) return BVH_E_INVALID_ARG;
    switch (option) {
        case BVH_OPT_HPLOC_SCHEDULER: if (value < 0 || value > 3) return BVH_E_INVALID_ARG; break;
        case BVH_OPT_LBVH_SCHEDULER: if (value < 0 || value > 2) return BVH_E_INVALID_ARG; break;
        case BVH_OPT_PLOC_SCHEDULER: if (value < 0 || value > 2) return BVH_E_INVALID_ARG; if (value == 2) value = 0; break;   // (2: ABI 4's cooperative launch, removed in round 5 — same trees; still accepted, means 0)
        case BVH_OPT_SORT_TEST_KNOBS: if (value & ~(int64_t)(8 | 32)) return BVH_E_INVALID_ARG; break;
        default: return BVH_E_INVALID_ARG;
    }
    c->options[(int)option] = value;
    c->sort.test_knobs = (int)c->options[BVH_OPT_SORT_TEST_KNOBS];
    c->ploc.static_ids = c->options[BVH_OPT_PLOC_SCHEDULER] == 0;
    return 0;
}
#ifdef BVH_ABLATION   // measurement build only: raw reads of the HPLOC queue buffers (the upper half of queue_rng doubles as a trace buffer: tools/ext_trace.py)
extern "C" int bvh_debug_read_queue(bvh_ctx* c, int which, size_t off_words, void* dst, size_t nwords) {
    if (!c || !dst) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return BVH_E_INTERNAL;
    if (which == 0) { if (hipMemcpy(dst, c->hploc.queue_count + off_words, nwords * 4, hipMemcpyDeviceToHost) != hipSuccess) return BVH_E_INTERNAL; }
    else if (which == 1) { if (off_words + nwords > c->hploc.queue_capacity || hipMemcpy(dst, c->hploc.queue_rng + off_words, nwords * 8, hipMemcpyDeviceToHost) != hipSuccess) return BVH_E_INTERNAL; }
    else if (which == 2) { *(uint64_t*)dst = c->hploc.queue_capacity; }
    else return BVH_E_INVALID_ARG;
    return 0;
}
#endif
int bvh_ctx_get_option(const bvh_ctx* c, bvh_option option, int64_t* value_out) {
#ifdef BVH_ABLATION   // measurement build only: option 1000 = merge tasks the HPLOC tile kernel ran in the last build (tools/measure_task_share.py)
    if (c && value_out && (int)option >= 1000 && (int)option < 1000 + 2046 + 32) {      // 1001..: the other measurement words of sub-queue 0's padded head (ABL_EXT_TIMING)
        u32 v = 0; Bind b(c->device);
        if (!c->hploc.queue_count || hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&v, c->hploc.queue_count + 2 + ((int)option - 1000), 4, hipMemcpyDeviceToHost) != hipSuccess) return BVH_E_INTERNAL;
        *value_out = v; return 0;
    }
#endif
    if (!c || !value_out || (int)option < 0 || (int)option > 3) return BVH_E_INVALID_ARG;
    *value_out = c->options[(int)option];
    return 0;
}

int bvh_ctx_create_on_stream(int device, void* hip_stream, bvh_ctx** out) {
    if (!out) return BVH_E_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return BVH_E_INVALID_ARG;
    Bind b(device); if (!b.ok) return BVH_E_INVALID_ARG;
    bvh_ctx* c = new (std::nothrow) bvh_ctx();
    if (!c) return BVH_E_INTERNAL;
    c->device = device;
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else { hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); if (e != hipSuccess) { c->stream = nullptr; bvh_ctx_destroy(c); return -(int)e; } c->own_stream = true; }
    // (a failure from here on goes through bvh_ctx_destroy, which releases whatever exists: stream, events, pinned words)
    for (auto& e : c->ev) { hipError_t r = hipEventCreate(&e); if (r != hipSuccess) { e = nullptr; bvh_ctx_destroy(c); return -(int)r; } }
    { hipError_t r = hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), (16 + PLOC_STATE_WORDS) * sizeof(u32), hipHostMallocMapped); if (r != hipSuccess) { c->h_pinned = nullptr; bvh_ctx_destroy(c); return -(int)r; }
      void* dp = nullptr; r = hipHostGetDevicePointer(&dp, c->h_pinned, 0); if (r != hipSuccess) { bvh_ctx_destroy(c); return -(int)r; } c->d_pinned = static_cast<u32*>(dp); }
    // the build path's code objects are loaded here, once per process and device, not by a context's first build (first build of a fresh process at 262 144 triangles:
    // 2.4 ms against 0.13 warm; first HPLOC / PLOC++ build after that 0.51 / 0.65 against 0.18 / 0.38 — tools/cold_probe.py)
    { static std::once_flag warmed[64];
      std::call_once(warmed[device & 63], [] { warm_stage_em(); warm_sort(); warm_lbvh(); warm_hploc(); warm_ploc(); warm_misc(); warm_collapse(); }); }
    *out = c;
    return 0;
}
int bvh_ctx_create(int device, bvh_ctx** out) { return bvh_ctx_create_on_stream(device, nullptr, out); }

void bvh_ctx_destroy(bvh_ctx* c) {
    if (!c) return;
    Bind b(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->arena) hipFree(c->arena);
    if (c->tris) hipFree(c->tris);
    for (auto& e : c->ev) if (e) hipEventDestroy(e);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int bvh_ctx_reserve(bvh_ctx* c, uint32_t n) { if (!c) return BVH_E_INVALID_ARG; Bind b(c->device); return ensure_capacity(c, n); }
int bvh_ctx_device(const bvh_ctx* c) { return c ? c->device : -1; }
void* bvh_ctx_stream(const bvh_ctx* c) { return c ? (void*)c->stream : nullptr; }
int bvh_ctx_set_profiling(bvh_ctx* c, int level) {
    if (!c) return BVH_E_INVALID_ARG;
    c->profiling = level != 0; c->kernel_profiling = level >= 2; c->recorder.reset(); c->build_counter = 0;   // (the next build is a sampled one)
    return 0;
}

// Sum of the per-kernel event times recorded since the last bvh_ctx_set_profiling(ctx, 2).  Synchronises the stream.
// names_out: buffer for '\n'-separated kernel names; ms_out/count_out: per distinct kernel (order of first launch).
int bvh_ctx_kernel_times(bvh_ctx* c, char* names_out, uint32_t names_cap, float* ms_out, uint32_t* count_out, uint32_t max_kernels) {
    if (!c) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<std::string> order; std::map<std::string, std::pair<double, uint32_t>> acc;
    for (size_t i = 0; i + 1 < c->recorder.used; ++i) {
        if (!c->recorder.recs[i].name) continue;              // end-of-build mark
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->recorder.recs[i].ev, c->recorder.recs[i + 1].ev));
        auto it = acc.find(c->recorder.recs[i].name);
        if (it == acc.end()) { order.push_back(c->recorder.recs[i].name); acc[c->recorder.recs[i].name] = {ms, 1u}; }
        else { it->second.first += ms; it->second.second += 1; }
    }
    std::string names; uint32_t k = 0;
    for (auto& nme : order) { if (k >= max_kernels) break; if (ms_out) ms_out[k] = (float)acc[nme].first; if (count_out) count_out[k] = acc[nme].second; names += nme; names += '\n'; ++k; }
    if (names_out && names_cap) { std::strncpy(names_out, names.c_str(), names_cap - 1); names_out[names_cap - 1] = 0; }
    return (int)k;
}
int bvh_ctx_last_collapse_ms(const bvh_ctx* c, float* ms_out) { if (!c || !ms_out) return BVH_E_INVALID_ARG; *ms_out = c->last_collapse_ms; return 0; }
int bvh_ctx_set_kernel_filter(bvh_ctx* c, const char* kernel_name) {
    if (!c) return BVH_E_INVALID_ARG;
    c->recorder.only = kernel_name ? kernel_name : ""; c->recorder.reset();
    return 0;
}
int bvh_ctx_set_kernel_sampling(bvh_ctx* c, uint32_t every) {
    if (!c || every == 0) return BVH_E_INVALID_ARG;
    c->sample_every = every; c->build_counter = 0; c->recorder.reset();
    return 0;
}
int bvh_ctx_synchronize(bvh_ctx* c) { if (!c) return BVH_E_INVALID_ARG; Bind b(c->device); return herr(hipStreamSynchronize(c->stream)); }

int bvh_stage_extents(bvh_ctx* c, const void* d_tris, uint32_t n, void* d_prim_aabbs, void* d_scene_extent) {
    if (!c || !d_tris || !d_prim_aabbs || !d_scene_extent || n == 0) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    launch_extents(c->stream, d_tris, n, d_prim_aabbs, d_scene_extent);
    return herr(hipGetLastError());
}

int bvh_stage_morton(bvh_ctx* c, const void* d_prim_aabbs, uint32_t n, const void* d_scene_extent, uint32_t* d_keys, uint32_t* d_vals) {
    if (!c || !d_prim_aabbs || !d_scene_extent || !d_keys || n == 0) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    launch_morton(c->stream, d_prim_aabbs, n, d_scene_extent, d_keys, d_vals, nullptr, 0, 0);
    return herr(hipGetLastError());
}

int bvh_sort_pairs(bvh_ctx* c, const uint32_t* d_keys_in, const uint32_t* d_vals_in, uint32_t n, uint32_t* d_keys_out,
                   uint32_t* d_vals_out, int start_bit, int end_bit) {
    if (!c || !d_keys_in || !d_keys_out || !d_vals_out || n == 0 || start_bit < 0 || end_bit > 32 || start_bit > end_bit) return BVH_E_INVALID_ARG;
    if (n >= (1u << 30)) return BVH_E_TOO_LARGE;
    Bind b(c->device);
    int r = ensure_capacity(c, n); if (r) return r;
    sort_prepare(c->stream, c->sort, n, sort_passes(start_bit, end_bit));
    sort_pairs(c->stream, c->sort, d_keys_in, d_vals_in, n, d_keys_out, d_vals_out, start_bit, end_bit, false);
    return herr(hipGetLastError());
}

int bvh_sort_pairs64(bvh_ctx* c, const uint64_t* d_keys_in, const uint32_t* d_vals_in, uint32_t n, uint64_t* d_keys_out,
                     uint32_t* d_vals_out, int start_bit, int end_bit) {
    if (!c || !d_keys_in || !d_keys_out || !d_vals_out || n == 0 || start_bit < 0 || end_bit > 64 || start_bit > end_bit) return BVH_E_INVALID_ARG;
    if (n >= (1u << 30)) return BVH_E_TOO_LARGE;
    Bind b(c->device);
    int r = ensure_capacity(c, n); if (r) return r;
    sort_prepare(c->stream, c->sort, n, sort_passes(start_bit, end_bit));
    sort_pairs64(c->stream, c->sort, d_keys_in, d_vals_in, n, d_keys_out, d_vals_out, start_bit, end_bit, false);
    return herr(hipGetLastError());
}

int bvh_stage_morton64(bvh_ctx* c, const void* d_prim_aabbs, uint32_t n, const void* d_scene_extent, uint64_t* d_keys, int total_bits) {
    if (!c || !d_prim_aabbs || !d_scene_extent || !d_keys || n == 0 || total_bits < 3 || total_bits > 60) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    launch_morton64(c->stream, d_prim_aabbs, n, d_scene_extent, d_keys, total_bits, nullptr, 0);
    return herr(hipGetLastError());
}

int bvh_stage_morton_plan(bvh_ctx* c, const void* d_scene_extent, int total_bits, int32_t plan_out[10]) {
    if (!c || !d_scene_extent || !plan_out || total_bits < 3 || total_bits > 60) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    int r = ensure_capacity(c, 2); if (r) return r;
    int* d = reinterpret_cast<int*>(c->small + 48);                      // 10 words of the ctx's small scratch
    launch_morton_plan(c->stream, d_scene_extent, d, total_bits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(plan_out, d, 10 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    return herr(hipStreamSynchronize(c->stream));
}

int bvh_stage_extents_ex(bvh_ctx* c, const bvh_build_input* in, uint32_t n, void* d_prim_aabbs, void* d_scene_extent) {
    if (!c || !in || !d_prim_aabbs || !d_scene_extent || n == 0) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    int r = stage_extents_fmt(c->stream, in, n, d_prim_aabbs, d_scene_extent); if (r) return r;
    return herr(hipGetLastError());
}

int bvh_emit_lbvh_single(bvh_ctx* c, const void* d_prim_aabbs, const uint32_t* d_sorted_keys, const uint32_t* d_sorted_vals,
                         uint32_t n, void* d_nodes, uint32_t* root_out) {
    if (!c || !d_prim_aabbs || !d_sorted_keys || !d_sorted_vals || !d_nodes || n < 2) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    int r = ensure_capacity(c, n); if (r) return r;
    r = begin_emit(c); if (r) return r;
    launch_lbvh_single(c->stream, d_prim_aabbs, d_sorted_keys, 32, d_sorted_vals, n, d_nodes, c->hploc.dep, c->small, c->ploc.list0, c->lbvh_queue_capacity, c->hploc.queue_count, false, (int)c->options[BVH_OPT_LBVH_SCHEDULER]);
    r = end_emit(c); if (r) return r;
    if (root_out) { r = read_back(c, c->small, 1, nullptr, 0, c->h_pinned); if (r) return r; *root_out = c->h_pinned[0]; }
    return 0;
}

int bvh_emit_lbvh_two(bvh_ctx* c, const void* d_prim_aabbs, const uint32_t* d_sorted_keys, const uint32_t* d_sorted_vals,
                      uint32_t n, void* d_nodes) {
    if (!c || !d_prim_aabbs || !d_sorted_keys || !d_sorted_vals || !d_nodes || n < 2) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    int r = ensure_capacity(c, n); if (r) return r;
    r = begin_emit(c); if (r) return r;
    launch_lbvh_two(c->stream, d_prim_aabbs, d_sorted_keys, 32, d_sorted_vals, n, d_nodes, c->parent, c->flags, c->hploc.dep, c->small,
                    c->ploc.list0, c->lbvh_queue_capacity, c->hploc.queue_count, false, (int)c->options[BVH_OPT_LBVH_SCHEDULER]);
    return end_emit(c);
}

int bvh_emit_ploc(bvh_ctx* c, const void* d_prim_aabbs, const uint32_t* d_sorted_vals, uint32_t n, void* d_nodes, void* d_leaves,
                  uint32_t* iterations_out) {
    if (!c || !d_prim_aabbs || !d_sorted_vals || !d_nodes || !d_leaves || n < 2) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    int r = ensure_capacity(c, n); if (r) return r;
    ploc_begin(c->stream, c->ploc, n);
    HIP_TRY(hipGetLastError());
    return run_ploc(c, n, d_nodes, d_leaves, d_prim_aabbs, d_sorted_vals, c->ploc, iterations_out);
}

int bvh_emit_hploc(bvh_ctx* c, const void* d_prim_aabbs, const uint32_t* d_sorted_keys, const uint32_t* d_sorted_vals, uint32_t n,
                   void* d_nodes, void* d_leaves) {
    if (!c || !d_prim_aabbs || !d_sorted_keys || !d_sorted_vals || !d_nodes || !d_leaves || n < 2) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    int r = ensure_capacity(c, n); if (r) return r;
    r = begin_emit(c); if (r) return r;
    emit_hploc(c, c->stream, d_prim_aabbs, d_sorted_keys, 32, d_sorted_vals, n, d_nodes, d_leaves);
    return end_emit(c);
}

#ifndef LEAF_FROM_TRIS
#define LEAF_FROM_TRIS 0
#endif
#ifndef SORT_GATE_TOP
#define SORT_GATE_TOP 1      // 0: the build sorts all 32 key bits at every size (no narrow top pass, no gate)
#endif
static int build_impl(bvh_ctx* c, bvh_algo algo, const bvh_build_input* in, uint32_t n, bvh_result* out, bvh_timings* tm) {
    const int key_bits = in->morton_bits == 60 ? 64 : 32;
    hipStream_t s = c->stream;
    const bool sampled = (c->build_counter++ % c->sample_every) == 0u;      // bvh_ctx_set_kernel_sampling: events in every n-th build only
    const bool prof = c->profiling && sampled;
    struct Install { bool on; explicit Install(bvh_ctx* c, bool sampled) : on(c->kernel_profiling && sampled) { if (on) g_recorder = &c->recorder; } ~Install() { if (on) g_recorder = nullptr; } } install(c, sampled);
    uint32_t ploc_iters = 0;
    int r = 0;
    if (prof) HIP_TRY(hipEventRecord(c->ev[0], s));
    // E: CalculateSceneExtents (token CalculateCentroidExtentsTime).  The sort's bookkeeping is cleared here so that the
    // Morton kernel can accumulate the digit histograms.
    // the reference sorts all 32 key bits at its four call sites (src/Hploc.cpp:63-81 ...); the codes this build produces have 30 (60) significant bits, so
    // the build's own sort runs on bits [0, 30) / [0, 60): four (eight) passes either way, but the last one has a 6-bit (4-bit) digit — 64 digit threads,
    // 6 ballots per key and 64 status words per tile in the look-back instead of 256.  The fused digit histograms of the Morton kernel
    // — (code >> 24) & 255 for the last pass — hold exactly those digits' counts.  (bvh_sort_pairs keeps full generality for caller-supplied keys.)
    // Round 5: the reference's wrap-around code arithmetic leaves bits 30 / 31 (60..63) set on degenerate scenes (planar, axis ratio >= 2^32).  The u64 sort
    // therefore runs on all 64 bits (eight passes either way); the u32 sort keeps its 6-bit top pass but gates it on a device word the Morton kernel raises when
    // such a code exists — the full-width instantiation enqueued behind it then does the pass (SORT_WIDE_FLAG_WORD, kernels.hpp).
    // Same-box A/B of the gate (gpurun_out/r5_gate_ab.log): + 3 us per build (the second launch) against the 4.5 us the narrow pass saves at 10 M — and nothing to save
    // below the wide sort tiles' threshold (262 144: 0.1141 -> 0.1165 ms), where the build therefore simply sorts all 32 bits like the reference.
    const int end_bit = key_bits == 64 ? 64 : (n >= SORT_WIDE_MIN_N && SORT_GATE_TOP != 0) ? 30 : 32;
    const int passes = sort_passes(0, end_bit);
    r = stage_extents_valid(in); if (r) return r;
    // what a build needs cleared (digit histograms, look-back status rows and tile tickets of `passes` sort passes, the emitters' queue heads) is cleared by
    // stage E's kernel itself; the scene extent of THIS build was reset by the previous build's Morton kernel (two extents, used alternately) — explicitly
    // only for a context's first build, after a re-allocation or after a build that failed half-way
    float* const scene = c->scene + 8 * c->scene_slot;
    float* const scene_next = c->scene + 8 * (c->scene_slot ^ 1);
    PrepArgs prep;
    prep.hist = c->sort.hist; prep.hist_words = (u32)SORT_HIST_COPIES * SORT_HIST_STRIDE;
    prep.status = reinterpret_cast<uint4*>(c->sort.status); prep.status_vecs = (u32)(((size_t)passes * sort_tiles(n) * SORT_RADIX) / 4);
    prep.counters = c->sort.counters; prep.extra = c->hploc.queue_count; prep.extra_words = hploc_head_words();
    if (algo == BVH_PLOCPP) ploc_begin_prep(c->ploc, n, prep);       // (the stage entry point bvh_emit_ploc launches k_ploc_init instead)
    const bool explicit_reset = !c->scene_ready;
    c->scene_ready = false;
    r = stage_extents_fmt(s, in, n, c->boxes, scene, explicit_reset, &prep); if (r) return r;
    if (prof) HIP_TRY(hipEventRecord(c->ev[1], s));
    // M: CalculateMortonCodes (token CalculateMortonCodesTime); values are implicit (value i = i), produced by sort pass 0
    if (key_bits == 64) launch_morton64(s, c->boxes, n, scene, reinterpret_cast<u64*>(c->keys), 60, c->sort.hist, passes, scene_next);
    else launch_morton(s, c->boxes, n, scene, c->keys, nullptr, c->sort.hist, SORT_BITS, passes, scene_next);
    if (prof) HIP_TRY(hipEventRecord(c->ev[2], s));
    // S: radix sort (token SortingTime)
    if (key_bits == 64) sort_pairs64(s, c->sort, reinterpret_cast<const u64*>(c->keys), nullptr, n, reinterpret_cast<u64*>(c->skeys), c->svals, 0, end_bit, true);
    else sort_pairs(s, c->sort, c->keys, nullptr, n, c->skeys, c->svals, 0, end_bit, true, end_bit == 30);
    if (prof) HIP_TRY(hipEventRecord(c->ev[3], s));
    // B: hierarchy emit (token BvhBuildTime; SetupClusters is booked here, not under Morton as the reference does)
    out->d_leaves = nullptr; out->layout = 0; out->root = 0;
    r = begin_emit(c); if (r) return r;
    switch (algo) {
        case BVH_LBVH_SINGLEPASS: launch_lbvh_single(s, c->boxes, c->skeys, key_bits, c->svals, n, c->nodes, c->hploc.dep, c->small, c->ploc.list0, c->lbvh_queue_capacity, c->hploc.queue_count, true, (int)c->options[BVH_OPT_LBVH_SCHEDULER]); break;
        case BVH_LBVH_TWOPASS:    launch_lbvh_two(s, c->boxes, c->skeys, key_bits, c->svals, n, c->nodes, c->parent, c->flags, c->hploc.dep, c->small,
                                                  c->ploc.list0, c->lbvh_queue_capacity, c->hploc.queue_count, true, (int)c->options[BVH_OPT_LBVH_SCHEDULER]); break;
        case BVH_HPLOC:           c->hploc.leaf_tris = (LEAF_FROM_TRIS && in->tri_format == BVH_TRI_PADDED64) ? in->d_tris : nullptr;
                                  emit_hploc(c, s, c->boxes, c->skeys, key_bits, c->svals, n, c->nodes, c->leaves, true);
                                  c->hploc.leaf_tris = nullptr;
                                  out->d_leaves = c->leaves; out->layout = 1; break;
        case BVH_PLOCPP:          r = run_ploc(c, n, c->nodes, c->leaves, c->boxes, c->svals, c->ploc, &ploc_iters); if (r) return r;
                                  out->d_leaves = c->leaves; out->layout = 1; break;
    }
    r = end_emit(c); if (r) return r;
    c->scene_ready = true; c->scene_slot ^= 1;             // every launch was accepted: the Morton kernel has reset the other extent for the next build
    if (install.on) c->recorder.mark(s, nullptr);
    if (prof) HIP_TRY(hipEventRecord(c->ev[4], s));
    if (algo == BVH_LBVH_SINGLEPASS) {   // m_rootNodeIdx read-back (src/SinglePassLbvh.cpp:131)
        r = read_back(c, c->small, 1, nullptr, 0, c->h_pinned); if (r) return r;           // {root, ~root} into pinned words 0..1
        out->root = c->h_pinned[0];
        if (out->root >= 2u * n - 1u) return BVH_E_INTERNAL;                               // (an index that consumers turn into an address)
    }
    out->d_nodes = c->nodes; out->d_prim_aabbs = c->boxes; out->d_scene_extent = scene;
    out->d_sorted_keys = c->skeys; out->d_sorted_vals = c->svals;
    out->n_internal = n - 1; out->n_leaves = n; out->key_bits = (uint32_t)key_bits; out->reserved = 0;
    out->d_tris = in->tri_format == BVH_TRI_INDEXED ? in->d_vertices : in->d_tris; out->d_morton_keys = c->keys;
    if (tm) {
        std::memset(tm, 0, sizeof *tm);
        tm->bytes_algorithmic = algorithmic_bytes(algo, n);
        tm->ploc_iterations = ploc_iters;
        tm->sampled = prof ? 1u : 0u;
        if (prof) {
            HIP_TRY(hipEventSynchronize(c->ev[4]));
            HIP_TRY(hipEventElapsedTime(&tm->ms_extents, c->ev[0], c->ev[1]));
            HIP_TRY(hipEventElapsedTime(&tm->ms_morton, c->ev[1], c->ev[2]));
            HIP_TRY(hipEventElapsedTime(&tm->ms_sort, c->ev[2], c->ev[3]));
            HIP_TRY(hipEventElapsedTime(&tm->ms_build, c->ev[3], c->ev[4]));
            tm->ms_total = tm->ms_extents + tm->ms_morton + tm->ms_sort + tm->ms_build;
        }
    }
    return 0;
}

int bvh_build(bvh_ctx* c, bvh_algo algo, const void* tris, uint32_t n, int tris_on_device, bvh_result* out, bvh_timings* tm) {
    if (!c || !tris || !out || n < 2 || (int)algo < 0 || (int)algo > 3) return BVH_E_INVALID_ARG;
    if (n >= (1u << 30)) return BVH_E_TOO_LARGE;
    Bind b(c->device);
    int r = ensure_capacity(c, n); if (r) return r;
    bvh_build_input in; std::memset(&in, 0, sizeof in);
    in.tri_format = BVH_TRI_PADDED64; in.morton_bits = 30; in.d_tris = tris;
    if (!tris_on_device) {   // H2D copy of the input, outside the timers (src/TwoPassLbvh.cpp:19-20)
        r = ensure_tris(c, n); if (r) return r;
        HIP_TRY(hipMemcpyAsync(c->tris, tris, (size_t)n * sizeof(bvh_triangle), hipMemcpyHostToDevice, c->stream));
        in.d_tris = c->tris;
    }
    return build_impl(c, algo, &in, n, out, tm);
}

int bvh_build_ex(bvh_ctx* c, bvh_algo algo, const bvh_build_input* in, uint32_t n, bvh_result* out, bvh_timings* tm) {
    if (!c || !in || !out || n < 2 || (int)algo < 0 || (int)algo > 3) return BVH_E_INVALID_ARG;
    if (in->morton_bits != 30 && in->morton_bits != 60) return BVH_E_INVALID_ARG;
    if (n >= (1u << 30)) return BVH_E_TOO_LARGE;
    Bind b(c->device);
    int r = ensure_capacity(c, n); if (r) return r;
    return build_impl(c, algo, in, n, out, tm);
}

int bvh_to_lbvh_layout(bvh_ctx* c, const bvh_result* in, void* d_out) {
    if (!c || !in || !d_out || !in->d_nodes) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    if (in->layout == 0) { HIP_TRY(hipMemcpyAsync(d_out, in->d_nodes, (size_t)(2 * in->n_leaves - 1) * sizeof(bvh2_node), hipMemcpyDeviceToDevice, c->stream)); return 0; }
    launch_to_lbvh_layout(c->stream, in->d_nodes, in->d_leaves, in->n_leaves, d_out);
    return herr(hipGetLastError());
}

// CollapseToWide4Bvh + its host set-up (src/TwoPassLbvh.cpp:154-183).  d_bvh4: Bvh4Node[n] (128 B), d_primnodes: PrimNode[n].
int bvh_collapse4(bvh_ctx* c, const bvh_result* in, void* d_bvh4, void* d_primnodes, uint32_t* n_wide_out) {
    if (!c || !in || !in->d_nodes || !d_bvh4 || !d_primnodes) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    const uint32_t n = in->n_leaves;
    int r = ensure_capacity(c, n); if (r) return r;
    hipStream_t s = c->stream;
    uint2* taskq = reinterpret_cast<uint2*>(c->slots);                 // u64[n] scratch, free after a build
    u32* state = c->ploc.state;                                         // >= COLLAPSE_STATE_WORDS words
    static_assert(PLOC_STATE_WORDS >= COLLAPSE_STATE_WORDS, "state scratch");
    if (c->profiling) HIP_TRY(hipEventRecord(c->ev[5], s));
    collapse_begin(s, taskq, state, in->root, true);
    // a batch's level counts come back into pinned words (a copy into pageable memory goes through a staging buffer and blocks); the host polls every one
    // of them (counts are < 2^31: the sentinel is not a count) instead of synchronising the stream — see wait_readback
    u32* const host_pairs = c->h_pinned + 16;
    static_assert(2 * COLLAPSE_MAX_BATCH <= PLOC_STATE_WORDS, "pinned read-back words");
    // wide levels ~ half the BVH2 depth: a first batch sized for a balanced tree — or one above what the previous collapse of a tree of this size needed
    // (animation frames, the benchmark loop: a level launched after the end costs ~3 us) —, then batches of 16 until a level creates nothing
    int batch = 10; for (uint32_t m = n; m > 1u; m >>= 1) batch += 1;
    const uint64_t hint_key = ((uint64_t)in->layout << 32) | (uint64_t)in->root;
    const bool hinted = c->collapse_last_n == n && c->collapse_last_key == hint_key;
    if (hinted && c->collapse_last_levels > 0) batch = (int)c->collapse_last_levels + 1;
    if (batch > COLLAPSE_MAX_BATCH) batch = COLLAPSE_MAX_BATCH;
    u32 base_begin = 0, base_len = 1, levels = 0;                      // the root task
    for (long long total = 0; total < (1ll << 31); total += batch, batch = 16) {
        const bool known = total == 0 && hinted && c->collapse_len_batch > 0 && batch <= c->collapse_len_batch + 1;
        collapse_enqueue(s, in->d_nodes, in->d_leaves, d_bvh4, d_primnodes, taskq, state, base_begin, base_len, batch, n, (int)in->layout,
                         known ? c->collapse_last_len : nullptr);
        HIP_TRY(hipGetLastError());
        r = read_back(c, state, (u32)batch, nullptr, 0, host_pairs); if (r) return r;
        u32 host[COLLAPSE_MAX_BATCH];
        for (int l = 0; l < batch; ++l) host[l] = host_pairs[2 * l];
        u32 len = base_len, allocated = base_begin + base_len;       // ids handed out so far
        u32 lens[COLLAPSE_MAX_BATCH + 1]; lens[0] = base_len;
        for (int l = 0; l < batch; ++l) { if (len) ++levels; len = host[l]; allocated += len; lens[l + 1] = len; }
        if (host[batch - 1] == 0u) {                   // the batch's last level allocated nothing: every later one would have no work
            if (n_wide_out) *n_wide_out = allocated;
            // (remembered only when the whole collapse was one batch: then level l of the next first batch is level l of this one)
            c->collapse_last_n = n; c->collapse_last_levels = levels; c->collapse_last_key = hint_key;
            c->collapse_len_batch = 0;
            if (total == 0) { c->collapse_len_batch = batch; for (int l = 0; l < COLLAPSE_MAX_BATCH; ++l) c->collapse_last_len[l] = l < batch ? lens[l] : 0u; }
            if (c->profiling) {   // token CollapseBvhTime (src/TwoPassLbvh.cpp:182), including this implementation's level read-backs
                HIP_TRY(hipEventRecord(c->ev[6], s)); HIP_TRY(hipEventSynchronize(c->ev[6]));
                HIP_TRY(hipEventElapsedTime(&c->last_collapse_ms, c->ev[5], c->ev[6]));
            }
            return 0;
        }
        base_len = host[batch - 1]; base_begin = allocated - base_len;
        collapse_begin(s, taskq, state, 0u, false);    // clear the counters for the next batch
    }
    return BVH_E_INTERNAL;
}

// GenerateRays (src/CommonBlocksKernel.h:432-463): camera (64-byte Camera record, host pointer) -> Ray[width*height] on the device
int bvh_generate_rays(bvh_ctx* c, const void* h_camera, void* d_rays, uint32_t width, uint32_t height) {
    if (!c || !h_camera || !d_rays || !width || !height) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    HIP_TRY(hipMemcpyAsync(c->small + 16, h_camera, 64, hipMemcpyHostToDevice, c->stream));
    launch_generate_rays(c->stream, c->small + 16, d_rays, width, height);
    return herr(hipGetLastError());
}
// BvhTraversalWhile (src/TraversalKernel.h:238-335) over an LBVH-layout node array; d_rgba (width*height*4 bytes) is cleared first
// (src/TwoPassLbvh.cpp:243: d_colorBuffer.reset()).
int bvh_trace_while(bvh_ctx* c, const void* d_rays, const void* d_tris, const void* d_nodes_lbvh, uint32_t root, uint32_t n_internal,
                    const void* h_transform, void* d_rgba, uint32_t width, uint32_t height) {
    if (!c || !d_rays || !d_tris || !d_nodes_lbvh || !h_transform || !d_rgba || !width || width != height) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    HIP_TRY(hipMemcpyAsync(c->small + 32, h_transform, 64, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_rgba, 0, (size_t)width * height * 4, c->stream));
    launch_trace_while(c->stream, d_rays, d_tris, d_nodes_lbvh, c->small + 32, d_rgba, root, width, height, n_internal);
    return herr(hipGetLastError());
}

// the reference's four traversal kernels behind one entry point (src/TraversalKernel.h:49-146, :148-236, :238-335, :337-451)
int bvh_trace(bvh_ctx* c, bvh_trace_kind kind, const void* d_rays, const void* d_tris, const void* d_nodes_lbvh, uint32_t root, uint32_t n_internal,
              const void* h_transform, void* d_rgba, uint32_t* d_ray_counter, uint32_t width, uint32_t height) {
    if (kind == BVH_TRACE_WHILE_WHILE) {
        int r = bvh_trace_while(c, d_rays, d_tris, d_nodes_lbvh, root, n_internal, h_transform, d_rgba, width, height);
        if (r == 0 && d_ray_counter) { Bind b(c->device); HIP_TRY(hipMemsetAsync(d_ray_counter, 0, (size_t)width * height * 4, c->stream)); }   // (that kernel does not count)
        return r;
    }
    if (!c || !d_rays || !d_tris || !d_nodes_lbvh || !h_transform || !d_rgba || !width || width != height) return BVH_E_INVALID_ARG;
    if (kind != BVH_TRACE_RESTART_TRAIL && kind != BVH_TRACE_IF_IF && kind != BVH_TRACE_SPECULATIVE_WHILE) return BVH_E_INVALID_ARG;
    if (kind == BVH_TRACE_RESTART_TRAIL && root != 0) return BVH_E_INVALID_ARG;       // a restart re-enters at node 0 (src/TraversalKernel.h:44)
    Bind b(c->device);
    HIP_TRY(hipMemcpyAsync(c->small + 32, h_transform, 64, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(d_rgba, 0, (size_t)width * height * 4, c->stream));
    launch_trace_kind(c->stream, (int)kind, d_rays, d_tris, d_nodes_lbvh, c->small + 32, d_rgba, d_ray_counter, root, width, height, n_internal);
    return herr(hipGetLastError());
}

int bvh_sah_cost(bvh_ctx* c, const bvh_result* in, double* cost_out) {
    if (!c || !in || !cost_out || !in->d_nodes) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    double* d = reinterpret_cast<double*>(c->small + 8);
    launch_sah_cost(c->stream, in->d_nodes, in->d_leaves, in->root, in->n_leaves, (int)in->layout, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(cost_out, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return herr(hipStreamSynchronize(c->stream));
}

int bvh_bvh4_cost(bvh_ctx* c, const void* d_bvh4, uint32_t n_wide, const void* d_primnodes, const void* d_prim_aabbs, uint32_t n, double* cost_out) {
    if (!c || !d_bvh4 || !d_primnodes || !d_prim_aabbs || !cost_out || n < 2 || n_wide == 0 || n_wide > n) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    int r = ensure_capacity(c, 2); if (r) return r;
    double* d = reinterpret_cast<double*>(c->small + 8);
    launch_bvh4_cost(c->stream, d_bvh4, n_wide, d_primnodes, d_prim_aabbs, n, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(cost_out, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    return herr(hipStreamSynchronize(c->stream));
}

int bvh_checksum(bvh_ctx* c, const bvh_result* in, uint64_t* checksum_out) {
    if (!c || !in || !checksum_out || !in->d_nodes || in->n_leaves < 2) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    int r = ensure_capacity(c, 2); if (r) return r;
    uint64_t* d = reinterpret_cast<uint64_t*>(c->small + 10);
    const uint32_t n = in->n_leaves;
    launch_checksum(c->stream, in->d_nodes, in->layout == 0 ? 2 * n - 1 : n - 1, in->layout == 1 ? in->d_leaves : nullptr, n, in->root, d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(checksum_out, d, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    return herr(hipStreamSynchronize(c->stream));
}

int bvh_download(bvh_ctx* c, const bvh_result* in, void* h_nodes, void* h_leaves, void* h_sorted_keys, uint32_t* h_sorted_vals, void* h_scene) {
    if (!c || !in) return BVH_E_INVALID_ARG;
    Bind b(c->device);
    const size_t n = in->n_leaves;
    const size_t node_count = in->layout == 0 ? 2 * n - 1 : n - 1;
    if (h_nodes) HIP_TRY(hipMemcpyAsync(h_nodes, in->d_nodes, node_count * sizeof(bvh2_node), hipMemcpyDeviceToHost, c->stream));
    if (h_leaves && in->d_leaves) HIP_TRY(hipMemcpyAsync(h_leaves, in->d_leaves, n * sizeof(bvh_primref), hipMemcpyDeviceToHost, c->stream));
    if (h_sorted_keys) HIP_TRY(hipMemcpyAsync(h_sorted_keys, in->d_sorted_keys, n * (in->key_bits == 64 ? 8 : 4), hipMemcpyDeviceToHost, c->stream));
    if (h_sorted_vals) HIP_TRY(hipMemcpyAsync(h_sorted_vals, in->d_sorted_vals, n * 4, hipMemcpyDeviceToHost, c->stream));
    if (h_scene) HIP_TRY(hipMemcpyAsync(h_scene, in->d_scene_extent, sizeof(bvh_aabb), hipMemcpyDeviceToHost, c->stream));
    return herr(hipStreamSynchronize(c->stream));
}

// plain device memory helpers so that non-HIP hosts (ctypes, cgo ...) can stage buffers without linking the HIP runtime
int bvh_dev_alloc(bvh_ctx* c, uint64_t bytes, void** out) { if (!c || !out) return BVH_E_INVALID_ARG; Bind b(c->device); return herr(hipMalloc(out, bytes)); }
int bvh_dev_free(bvh_ctx* c, void* p) { if (!c) return BVH_E_INVALID_ARG; Bind b(c->device); return herr(hipFree(p)); }
int bvh_dev_upload(bvh_ctx* c, void* d_dst, const void* h_src, uint64_t bytes) {
    if (!c) return BVH_E_INVALID_ARG; Bind b(c->device);
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream)); return herr(hipStreamSynchronize(c->stream)); }
int bvh_dev_download(bvh_ctx* c, void* h_dst, const void* d_src, uint64_t bytes) {
    if (!c) return BVH_E_INVALID_ARG; Bind b(c->device);
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream)); return herr(hipStreamSynchronize(c->stream)); }

int bvh_dev_copy(bvh_ctx* c, void* d_dst, const void* d_src, uint64_t bytes) {   // asynchronous, ordered on the ctx's stream; a kernel, not a runtime copy (misc.hip)
    if (!c || (bytes && (!d_dst || !d_src))) return BVH_E_INVALID_ARG; Bind b(c->device);
    launch_copy_bytes(c->stream, d_dst, d_src, (size_t)bytes);
    return herr(hipGetLastError()); }

} // extern "C"
