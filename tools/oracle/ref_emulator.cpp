// ref_emulator.cpp — runs the REFERENCE's own device code on the CPU under a small SIMT emulator: the kernel headers are included from where they
// lie under /root/reference (by path, never copied into the repository).  TEST INFRASTRUCTURE ONLY (dev container; the reference tree does not
// exist on the GPU box).  Compiled three times by oracle/Makefile:
//   -DREF_FLAVOUR_PLOC  src/Ploc++Kernel.h: SetupClusters, Ploc, SinglePassPloc driven by a restatement of the host loop (src/PLOC++Bvh.cpp:82-152) —
//                       the wave32 flavour (src/Common.h:100-106 without a gfx9 macro); since round 5 the reference's own wave64 flavour of the same kernels also
//                       runs on the MI355X (oracle/_ref/*.w64.co, tests/test_reference_w64.py), so this is a second, independent reading — and its
//                       CollapseToWide4Bvh (:364-465; host set-up src/PLOC++Bvh.cpp:154-190);
//   -DREF_FLAVOUR_LBVH  src/TwoPassLbvhKernel.h: CollapseToWide4Bvh (:237-336; host set-up src/TwoPassLbvh.cpp:154-183).
//   -DREF_FLAVOUR_HPLOC src/HplocKernel.h: SetupClusters + HPloc (host: src/Hploc.cpp:83-121).  This flavour exists to CHECK THE EMULATOR, not the oracle: the
//                       same header also runs UNMODIFIED on the MI355X (oracle/_ref/HplocKernel*.co through oracle/ref_driver.cpp), and it speaks the
//                       vocabulary the Ploc / SinglePassPloc kernels speak (__ballot, __shfl, __syncthreads, LDS atomicMin(u64), global atomicAdd / atomicExch).
//                       tests/test_reference_kernels.py::test_emulator_matches_hardware_on_hploc compares emulator(HplocKernel.h) with hardware(HplocKernel.h)
//                       on the golden meshes — topology hash, leaves, merged count — which validates the emulator's wave-operation / barrier / atomic
//                       semantics on silicon and thereby the instrument behind the PLOC++ and collapse pins.  Compiled from a temporary copy with the three
//                       probe-only edits of SURVEY.md Appendix A.3 (lock-step made explicit: a barrier after :116, the clear of :143 moved behind :181).
// tools/make_golden.py writes what they compute for the golden meshes into tests/golden/reference_outputs.json, which pins the oracle's PLOC++
// and collapse restatements.  Recipe: SURVEY.md Appendix A.3.
//
// The emulator (own code): one ucontext fiber per thread; __syncthreads, __threadfence and the wave operations (__ballot, __any, __shfl) are yield
// points; a 32-lane wave's operation resolves among the lanes that wait in it after every runnable thread has run up to its next yield point
// (divergent lanes simply are not among them); a barrier releases when only barrier waiters are left.  Two launch modes: workgroups one after
// another in blockIdx order (satisfies the Ploc kernel's spin on atomicBlockCounter, :341-347), or the whole grid resident at once (the collapse
// kernels spin until their task appears and hang unless every thread is resident, SURVEY.md Appendix B: here every thread is a fiber).  Atomics are
// plain read-modify-writes (fibers are cooperative).  Ploc++Kernel.h is compiled from a temporary copy with ONE inserted line — a __syncthreads()
// after line 187, where SinglePassPloc's scan scratch aliases nodeIndicesSharedMem and is overwritten by lanes that finished the scan while others
// still read it (lock-step execution hides that on the GPU) — made by oracle/Makefile with sed into $TMPDIR.
#include <ucontext.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <src/Common.h>

// ---- SIMT emulator ------------------------------------------------------------------------------------------------------
namespace emu {
enum State { RUN, BARRIER, WARPOP, DONE };
enum Op { OP_BALLOT, OP_SHFL };
struct Dim { unsigned x, y, z; };
struct Lane { ucontext_t ctx; State state; Dim tid, bid; Op op; long long val; int src; unsigned long long res; };
static Lane* g_cur = nullptr;
static ucontext_t g_sched;
static Dim g_bdim = {1, 1, 1};
static void (*g_entry)() = nullptr;
constexpr int kWave = 32;                                  // the reference's WarpSize on this target (src/Common.h:100-106)
constexpr size_t kStack = 128 * 1024;

inline void yield() { swapcontext(&g_cur->ctx, &g_sched); }
static void trampoline() { g_entry(); g_cur->state = DONE; yield(); }
inline unsigned long long warpop(Op op, long long val, int src) { g_cur->state = WARPOP; g_cur->op = op; g_cur->val = val; g_cur->src = src; yield(); return g_cur->res; }

// run workgroups [first_block, first_block + n_blocks) of `threads` threads each to completion, all of them resident together
static void run_blocks(unsigned first_block, unsigned n_blocks, unsigned threads, void (*entry)(), std::vector<Lane>& lanes, std::vector<char>& stacks) {
    g_bdim = {threads, 1, 1}; g_entry = entry;
    const size_t total = (size_t)n_blocks * threads;
    if (lanes.size() < total) { lanes.resize(total); stacks.resize(total * kStack); }
    for (size_t t = 0; t < total; ++t) {
        Lane& l = lanes[t];
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = stacks.data() + t * kStack; l.ctx.uc_stack.ss_size = kStack; l.ctx.uc_link = &g_sched;
        makecontext(&l.ctx, trampoline, 0);
        l.state = RUN; l.tid = {(unsigned)(t % threads), 0, 0}; l.bid = {first_block + (unsigned)(t / threads), 0, 0};
    }
    while (true) {
        // every runnable thread runs up to its next yield point (a thread that merely yielded — __threadfence in a spin loop — stays runnable)
        bool any_run = false;
        for (size_t t = 0; t < total; ++t) if (lanes[t].state == RUN) { g_cur = &lanes[t]; swapcontext(&g_sched, &lanes[t].ctx); any_run = true; }
        bool any_warp = false, any_barrier = false, any_live = false;
        for (size_t w = 0; w < total; w += kWave) {              // (threads per workgroup are multiples of 32 here, so waves do not straddle workgroups)
            const size_t e = std::min(total, w + kWave);
            bool have = false; Op op = OP_BALLOT;
            for (size_t t = w; t < e; ++t) if (lanes[t].state == WARPOP) { if (have && lanes[t].op != op) { fprintf(stderr, "emulator: mixed wave operations\n"); abort(); } have = true; op = lanes[t].op; }
            if (!have) continue;
            any_warp = true;
            if (op == OP_BALLOT) {
                unsigned long long mask = 0;
                for (size_t t = w; t < e; ++t) if (lanes[t].state == WARPOP && lanes[t].val) mask |= 1ull << (t - w);
                for (size_t t = w; t < e; ++t) if (lanes[t].state == WARPOP) { lanes[t].res = mask; lanes[t].state = RUN; }
            } else {
                long long vals[kWave]; bool in[kWave];
                for (int k = 0; k < kWave; ++k) { const size_t t = w + k; in[k] = t < e && lanes[t].state == WARPOP; vals[k] = in[k] ? lanes[t].val : 0; }
                for (int k = 0; k < kWave; ++k) if (in[k]) { Lane& l = lanes[w + k]; const int sl = l.src & (kWave - 1); l.res = (unsigned long long)(in[sl] ? vals[sl] : vals[k]); l.state = RUN; }
            }
        }
        if (any_warp) continue;
        for (size_t t = 0; t < total; ++t) { if (lanes[t].state == RUN) any_live = true; }
        if (any_live) continue;                                  // spinners: go round again
        // nobody runnable, no wave operation pending: release the barriers, workgroup by workgroup (all waiters of a workgroup together)
        for (size_t t = 0; t < total; ++t) if (lanes[t].state == BARRIER) { lanes[t].state = RUN; any_barrier = true; }
        if (!any_barrier) return;                                // everybody DONE
        (void)any_run;
    }
}
}  // namespace emu

// ---- the device-side vocabulary the kernel header uses --------------------------------------------------------------------
#define __global__
#define __shared__ static
#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_cur->bid)
#define blockDim (emu::g_bdim)
namespace BvhConstruction { constexpr int WarpSize = emu::kWave; }
inline void __syncthreads() { emu::g_cur->state = emu::BARRIER; emu::yield(); }
inline void __threadfence() { emu::yield(); }        // a yield point: spin loops around it let the other fibers run
inline uint64_t __ballot(int pred) { return emu::warpop(emu::OP_BALLOT, pred, 0); }
inline int __shfl(int v, int src) { return (int)emu::warpop(emu::OP_SHFL, v, src); }
inline int __any(int pred) { return __ballot(pred) != 0; }      // (the collapse kernel of the same header; never launched here)
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
template <typename T, typename U> inline T atomicExch(T* p, U v) { const T old = *p; *p = (T)v; return old; }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
template <typename T, typename U> inline T atomicAdd(T* p, U v) { const T old = *p; *p = (T)(old + (T)v); return old; }
template <typename T, typename U> inline T atomicMin(T* p, U v) { const T old = *p; if ((T)v < old) *p = (T)v; return old; }
// HIP resolves min/max of two different arithmetic types through a double-promoting overload (__clang_hip_cmath.h); same types: the type
template <typename T> inline T min(T a, T b) { return b < a ? b : a; }
template <typename T> inline T max(T a, T b) { return a < b ? b : a; }
template <typename A, typename B> inline double min(A a, B b) { return (double)b < (double)a ? (double)b : (double)a; }
template <typename A, typename B> inline double max(A a, B b) { return (double)a < (double)b ? (double)b : (double)a; }

#include REF_KERNEL_HEADER           // REF_FLAVOUR_PLOC: temporary copy of src/Ploc++Kernel.h with the one barrier; REF_FLAVOUR_LBVH: src/TwoPassLbvhKernel.h as it is

// ---- kernel launches: arguments travel through globals, every fiber calls the kernel with them ------------------------------
namespace {
std::vector<emu::Lane> g_lanes; std::vector<char> g_stacks;
// Kernel::launch(workSize, blockSize): ceil(work / block) workgroups; sequential: one workgroup at a time in blockIdx order
void launch_sequential(void (*entry)(), u32 work, u32 block) { for (u32 b = 0; b < (work + block - 1) / block; ++b) emu::run_blocks(b, 1, block, entry, g_lanes, g_stacks); }
void launch_resident(void (*entry)(), u32 work, u32 block) { emu::run_blocks(0, (work + block - 1) / block, block, entry, g_lanes, g_stacks); }

// CollapseToWide4Bvh's host set-up (src/TwoPassLbvh.cpp:154-183 = src/PLOC++Bvh.cpp:154-190): task queue all-invalid except the root task,
// internal-node offset 1, launch over ceil(2 n / 3) threads
struct CollapseArgs { Bvh2Node* nodes; PrimRef* leaves; Bvh4Node* wide; PrimNode* prims; uint2* taskq; u32* count; u32* offset; u32 ni, n; } g_c;
int run_collapse(void (*entry)(), void* nodes, void* leaves, u32 root, u32 n, void* wide_out, void* prims_out, u32* n_wide_out) {
    if (n < 2) return -1;
    std::vector<uint2> taskq(n, uint2{INVALID_NODE_IDX, INVALID_NODE_IDX});
    taskq[0] = uint2{root, INVALID_NODE_IDX};
    u32 count = 0, offset = 1;
    Bvh4Node* w = (Bvh4Node*)wide_out; PrimNode* p = (PrimNode*)prims_out;
    for (u32 i = 0; i < n; ++i) { w[i] = Bvh4Node(); p[i] = PrimNode(); }          // d_wideBvhNodes / d_wideLeafNodes start default-constructed (reset)
    g_c = { (Bvh2Node*)nodes, (PrimRef*)leaves, w, p, taskq.data(), &count, &offset, n - 1, n };
    launch_resident(entry, (2 * n + 2) / 3, 256);
    if (n_wide_out) *n_wide_out = offset;
    return count == n ? 0 : -2;
}
}  // namespace

#ifdef REF_FLAVOUR_PLOC
namespace {
struct Args { int* idx0; int* idx1; Bvh2Node* nodes; PrimRef* leaves; u32* svals; Aabb* boxes; int* merged; int* offsum; int* counter; u32 count; u32 ni; } g_a;
void entry_setup() { SetupClusters(g_a.nodes, g_a.leaves, g_a.svals, g_a.boxes, g_a.idx0, g_a.count); }
void entry_ploc() { Ploc(g_a.idx0, g_a.idx1, g_a.nodes, g_a.leaves, g_a.merged, g_a.offsum, g_a.counter, g_a.count, g_a.ni); }
void entry_single() { SinglePassPloc(g_a.idx0, g_a.nodes, g_a.leaves, g_a.count, g_a.ni); }
void entry_collapse() { CollapseToWide4Bvh(g_c.nodes, g_c.leaves, g_c.wide, g_c.prims, g_c.taskq, g_c.count, g_c.offset, g_c.ni, g_c.n); }
}  // namespace

// PLOCNew::build from SetupClusters on (src/PLOC++Bvh.cpp:82-152).  boxes: Aabb[n] by primitive index; sorted_vals: u32[n];
// nodes_out: Bvh2Node[n-1]; leaves_out: PrimRef[n]; *iterations_out: passes of the host loop (Ploc launches + the SinglePassPloc one)
extern "C" int ref_emu_ploc(const void* boxes, const uint32_t* sorted_vals, uint32_t n, void* nodes_out, void* leaves_out, uint32_t* iterations_out) {
    if (n < 2) return -1;
    const u32 ni = n - 1;
    std::vector<int> idx0(n, (int)INVALID_NODE_IDX), idx1(n, (int)INVALID_NODE_IDX);
    int merged = 0, offsum = 0, counter = 0;
    g_a = { idx0.data(), idx1.data(), (Bvh2Node*)nodes_out, (PrimRef*)leaves_out, const_cast<u32*>(sorted_vals), (Aabb*)const_cast<void*>(boxes), &merged, &offsum, &counter, n, ni };
    launch_sequential(entry_setup, n, 256);                    // setupClusterKernel.launch(primitiveCount)
    bool swap = false; u32 c = n, iters = 0;
    while (c > 1) {                                             // :132-152
        merged = offsum = counter = 0;
        g_a.idx0 = !swap ? idx0.data() : idx1.data(); g_a.idx1 = !swap ? idx1.data() : idx0.data(); g_a.count = c;
        ++iters;
        if (c < (u32)PlocBlockSize) { launch_sequential(entry_single, c, PlocBlockSize); break; }
        launch_sequential(entry_ploc, c, PlocBlockSize);
        c -= (u32)merged;
        swap = !swap;
    }
    if (iterations_out) *iterations_out = iters;
    return 0;
}
// CollapseToWide4Bvh of the PLOC layout (src/Ploc++Kernel.h:364-465).  nodes: Bvh2Node[>= 2n] — the kernel reads bvh2Nodes[leafIdx].m_aabb for leaf
// children (:395-396, values unused), i.e. beyond the n-1 internal nodes: the caller passes a padded copy.  wide_out: Bvh4Node[n], prims_out: PrimNode[n]
extern "C" int ref_emu_collapse_ploc(void* nodes_padded, void* leaves, uint32_t root, uint32_t n, void* wide_out, void* prims_out, uint32_t* n_wide_out) {
    return run_collapse(entry_collapse, nodes_padded, leaves, root, n, wide_out, prims_out, n_wide_out);
}
#endif

#ifdef REF_FLAVOUR_LBVH
namespace { void entry_collapse() { CollapseToWide4Bvh(g_c.nodes, g_c.wide, g_c.prims, g_c.taskq, g_c.count, g_c.offset, g_c.ni, g_c.n); } }
// CollapseToWide4Bvh of the LBVH layout (src/TwoPassLbvhKernel.h:237-336).  nodes: Bvh2Node[2n-1]
extern "C" int ref_emu_collapse_lbvh(void* nodes, uint32_t root, uint32_t n, void* wide_out, void* prims_out, uint32_t* n_wide_out) {
    return run_collapse(entry_collapse, nodes, nullptr, root, n, wide_out, prims_out, n_wide_out);
}
#endif

#ifdef REF_FLAVOUR_HPLOC
namespace {
struct HArgs { Bvh2Node* nodes; PrimRef* leaves; u32* svals; Aabb* boxes; u32* skeys; u32* idx; u32* parent; u32* merged; u32 n; } g_h;
void entry_hsetup() { SetupClusters(g_h.nodes, g_h.leaves, g_h.svals, g_h.boxes, g_h.idx, g_h.parent, g_h.n); }
void entry_hploc() { HPloc(g_h.nodes, g_h.leaves, g_h.skeys, g_h.idx, g_h.parent, g_h.merged, g_h.n, g_h.n - 1); }
}  // namespace
// HPLOC::build from SetupClusters on (src/Hploc.cpp:83-121): nodeIdx0 / parentIdx all-invalid, SetupClusters over n threads, HPloc over ceil((n-1) / 32)
// workgroups of 32 threads (cover_all: ceil(n / 32), so that leaf n-1 gets a thread when (n-1) % 32 == 0 — the reference under-launches).
// Workgroups run one after another: a walker that arrives first at a node retires (atomicExch hand-off, :278-295), nobody waits for anybody.
extern "C" int ref_emu_hploc(const void* boxes, const uint32_t* sorted_keys, const uint32_t* sorted_vals, uint32_t n, void* nodes_out, void* leaves_out,
                             uint32_t* merged_out, int cover_all) {
    if (n < 2) return -1;
    std::vector<u32> idx(n, INVALID_NODE_IDX), parent(n, INVALID_NODE_IDX);
    u32 merged = 0;
    g_h = { (Bvh2Node*)nodes_out, (PrimRef*)leaves_out, const_cast<u32*>(sorted_vals), (Aabb*)const_cast<void*>(boxes), const_cast<u32*>(sorted_keys),
            idx.data(), parent.data(), &merged, n };
    launch_sequential(entry_hsetup, n, 256);
    launch_sequential(entry_hploc, cover_all ? n : n - 1, HPlocBlockSize);
    if (merged_out) *merged_out = merged;
    return 0;
}
#endif
