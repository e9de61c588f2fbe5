// ploc_emulator.cpp — runs the REFERENCE's own PLOC++ device code (SetupClusters, Ploc, SinglePassPloc of
// /root/reference/src/Ploc++Kernel.h, included by path, never copied into the repository) on the CPU under a small SIMT emulator,
// driven by a restatement of the reference's host loop (src/PLOC++Bvh.cpp:82-152).  TEST INFRASTRUCTURE ONLY (dev container; the
// reference tree does not exist on the GPU box): tools/make_golden.py uses it to write the reference's topology hash / SAH /
// iteration count for the golden meshes into tests/golden/reference_outputs.json, which pins the oracle's PLOC++ restatement —
// in particular its multi-chunk path (n >= 1024), which the reference kernel cannot run on wave64 hardware (WarpSize is hard-coded
// to 32 for gfx950, src/Common.h:100-106).  Recipe: SURVEY.md Appendix A.3.
//
// The emulator (own code): one ucontext fiber per thread of a workgroup; __syncthreads and the wave operations (__ballot, __shfl)
// are yield points; a 32-lane wave's operation resolves among the lanes that wait in it once every thread of the workgroup is blocked
// (divergent lanes simply are not among them); a barrier releases when only barrier waiters are left; workgroups run one after another
// in blockIdx order (which satisfies the kernel's spin on atomicBlockCounter, :341-347).  Atomics are plain read-modify-writes (fibers
// are cooperative).  The header is compiled from a temporary copy with ONE inserted line — a __syncthreads() after line 187, where
// SinglePassPloc's scan scratch aliases nodeIndicesSharedMem and is overwritten by lanes that finished the scan while others still
// read it (lock-step execution hides that on the GPU) — made by oracle/Makefile with sed into $TMPDIR.
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
struct Lane { ucontext_t ctx; State state; Dim tid; Op op; long long val; int src; unsigned long long res; };
static Lane* g_cur = nullptr;
static ucontext_t g_sched;
static Dim g_block = {0, 1, 1}, g_bdim = {1, 1, 1};
static void (*g_entry)() = nullptr;
constexpr int kWave = 32;                                  // the reference's WarpSize on this target (src/Common.h:100-106)
constexpr size_t kStack = 128 * 1024;

inline void yield() { swapcontext(&g_cur->ctx, &g_sched); }
static void trampoline() { g_entry(); g_cur->state = DONE; yield(); }
inline unsigned long long warpop(Op op, long long val, int src) { g_cur->state = WARPOP; g_cur->op = op; g_cur->val = val; g_cur->src = src; yield(); return g_cur->res; }

// run one workgroup of `threads` threads to completion
static void run_block(unsigned block, unsigned threads, void (*entry)(), std::vector<Lane>& lanes, std::vector<char>& stacks) {
    g_block = {block, 0, 0}; g_bdim = {threads, 1, 1}; g_entry = entry;
    for (unsigned t = 0; t < threads; ++t) {
        Lane& l = lanes[t];
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = stacks.data() + (size_t)t * kStack; l.ctx.uc_stack.ss_size = kStack; l.ctx.uc_link = &g_sched;
        makecontext(&l.ctx, trampoline, 0);
        l.state = RUN; l.tid = {t, 0, 0};
    }
    while (true) {
        bool ran = false;
        for (unsigned t = 0; t < threads; ++t) if (lanes[t].state == RUN) { g_cur = &lanes[t]; swapcontext(&g_sched, &lanes[t].ctx); ran = true; }
        if (ran) continue;
        bool any_warp = false, any_barrier = false;
        for (unsigned w = 0; w < threads; w += kWave) {
            bool have = false; Op op = OP_BALLOT;
            for (unsigned t = w; t < std::min(threads, w + kWave); ++t) if (lanes[t].state == WARPOP) { if (have && lanes[t].op != op) { fprintf(stderr, "emulator: mixed wave operations\n"); abort(); } have = true; op = lanes[t].op; }
            if (!have) continue;
            any_warp = true;
            if (op == OP_BALLOT) {
                unsigned long long mask = 0;
                for (unsigned t = w; t < std::min(threads, w + kWave); ++t) if (lanes[t].state == WARPOP && lanes[t].val) mask |= 1ull << (t - w);
                for (unsigned t = w; t < std::min(threads, w + kWave); ++t) if (lanes[t].state == WARPOP) { lanes[t].res = mask; lanes[t].state = RUN; }
            } else {
                long long vals[kWave]; bool in[kWave];
                for (int k = 0; k < kWave; ++k) { const unsigned t = w + k; in[k] = t < threads && lanes[t].state == WARPOP; vals[k] = in[k] ? lanes[t].val : 0; }
                for (int k = 0; k < kWave; ++k) if (in[k]) { Lane& l = lanes[w + k]; const int s = l.src & (kWave - 1); l.res = (unsigned long long)(in[s] ? vals[s] : vals[k]); l.state = RUN; }
            }
        }
        if (any_warp) continue;
        for (unsigned t = 0; t < threads; ++t) if (lanes[t].state == BARRIER) { lanes[t].state = RUN; any_barrier = true; }
        if (!any_barrier) return;                              // everybody DONE
    }
}
}  // namespace emu

// ---- the device-side vocabulary the kernel header uses --------------------------------------------------------------------
#define __global__
#define __shared__ static
#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_block)
#define blockDim (emu::g_bdim)
namespace BvhConstruction { constexpr int WarpSize = emu::kWave; }
inline void __syncthreads() { emu::g_cur->state = emu::BARRIER; emu::yield(); }
inline void __threadfence() {}
inline uint64_t __ballot(int pred) { return emu::warpop(emu::OP_BALLOT, pred, 0); }
inline int __shfl(int v, int src) { return (int)emu::warpop(emu::OP_SHFL, v, src); }
inline int __any(int pred) { return __ballot(pred) != 0; }      // (the collapse kernel of the same header; never launched here)
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
template <typename T, typename U> inline T atomicAdd(T* p, U v) { const T old = *p; *p = (T)(old + (T)v); return old; }
template <typename T, typename U> inline T atomicMin(T* p, U v) { const T old = *p; if ((T)v < old) *p = (T)v; return old; }
// HIP resolves min/max of two different arithmetic types through a double-promoting overload (__clang_hip_cmath.h); same types: the type
template <typename T> inline T min(T a, T b) { return b < a ? b : a; }
template <typename T> inline T max(T a, T b) { return a < b ? b : a; }
template <typename A, typename B> inline double min(A a, B b) { return (double)b < (double)a ? (double)b : (double)a; }
template <typename A, typename B> inline double max(A a, B b) { return (double)a < (double)b ? (double)b : (double)a; }

#include PLOC_KERNEL_HEADER          // the reference's src/Ploc++Kernel.h (temporary copy with the one barrier; see the header comment)

// ---- kernel launches: arguments travel through globals, every fiber calls the kernel with them ------------------------------
namespace {
struct Args { int* idx0; int* idx1; Bvh2Node* nodes; PrimRef* leaves; u32* svals; Aabb* boxes; int* merged; int* offsum; int* counter; u32 count; u32 ni; } g_a;
void entry_setup() { SetupClusters(g_a.nodes, g_a.leaves, g_a.svals, g_a.boxes, g_a.idx0, g_a.count); }
void entry_ploc() { Ploc(g_a.idx0, g_a.idx1, g_a.nodes, g_a.leaves, g_a.merged, g_a.offsum, g_a.counter, g_a.count, g_a.ni); }
void entry_single() { SinglePassPloc(g_a.idx0, g_a.nodes, g_a.leaves, g_a.count, g_a.ni); }
void launch(void (*entry)(), u32 work, u32 block) {             // Kernel::launch(workSize, blockSize): ceil(work / block) workgroups
    static std::vector<emu::Lane> lanes; static std::vector<char> stacks;
    if (lanes.size() < block) { lanes.resize(block); stacks.resize((size_t)block * emu::kStack); }
    for (u32 b = 0; b < (work + block - 1) / block; ++b) emu::run_block(b, block, entry, lanes, stacks);
}
}  // namespace

// PLOCNew::build from SetupClusters on (src/PLOC++Bvh.cpp:82-152).  boxes: Aabb[n] by primitive index; sorted_vals: u32[n];
// nodes_out: Bvh2Node[n-1]; leaves_out: PrimRef[n]; *iterations_out: passes of the host loop (Ploc launches + the SinglePassPloc one)
extern "C" int ref_emu_ploc(const void* boxes, const uint32_t* sorted_vals, uint32_t n, void* nodes_out, void* leaves_out, uint32_t* iterations_out) {
    if (n < 2) return -1;
    const u32 ni = n - 1;
    std::vector<int> idx0(n, (int)INVALID_NODE_IDX), idx1(n, (int)INVALID_NODE_IDX);
    int merged = 0, offsum = 0, counter = 0;
    g_a = { idx0.data(), idx1.data(), (Bvh2Node*)nodes_out, (PrimRef*)leaves_out, const_cast<u32*>(sorted_vals), (Aabb*)const_cast<void*>(boxes), &merged, &offsum, &counter, n, ni };
    launch(entry_setup, n, 256);                               // setupClusterKernel.launch(primitiveCount): default block size
    bool swap = false; u32 c = n, iters = 0;
    while (c > 1) {                                             // :132-152
        merged = offsum = counter = 0;
        g_a.idx0 = !swap ? idx0.data() : idx1.data(); g_a.idx1 = !swap ? idx1.data() : idx0.data(); g_a.count = c;
        ++iters;
        if (c < (u32)PlocBlockSize) { launch(entry_single, c, PlocBlockSize); break; }
        launch(entry_ploc, c, PlocBlockSize);
        c -= (u32)merged;
        swap = !swap;
    }
    if (iterations_out) *iterations_out = iters;
    return 0;
}
