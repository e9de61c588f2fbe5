// graph_gap.hip — what does a kernel boundary cost on this box: N dependent launches of a tiny kernel enqueued one by one on a stream, against the same N kernel nodes
// replayed as ONE hipGraph (captured from the stream)?  Two kernel shapes: one 64-thread workgroup, and 256 workgroups x 1024 threads with 40 KB of LDS (a PLOC++ iteration's shape).
// hipcc --offload-arch=gfx950 -O2 tools/probes/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_small(unsigned* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1u; }
__global__ __launch_bounds__(1024) void k_big(unsigned* p) { __shared__ unsigned s[10240]; s[threadIdx.x] = p[blockIdx.x]; __syncthreads(); if (threadIdx.x == 0) p[blockIdx.x] = s[1023 - threadIdx.x % 7] + 1u; }
int main() {
    unsigned* d; CK(hipMalloc(&d, 4096 * 4)); CK(hipMemset(d, 0, 4096 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int shape = 0; shape < 2; ++shape) {
        for (int n : {8, 32, 128}) {
            auto enqueue = [&]() { for (int i = 0; i < n; ++i) { if (shape == 0) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d); else hipLaunchKernelGGL(k_big, dim3(256), dim3(1024), 0, s, d); } };
            for (int w = 0; w < 3; ++w) enqueue();
            CK(hipStreamSynchronize(s));
            float best_stream = 1e9f, best_graph = 1e9f;
            for (int rep = 0; rep < 10; ++rep) {
                CK(hipEventRecord(e0, s)); enqueue(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_stream) best_stream = ms;
            }
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal)); enqueue(); CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            for (int rep = 0; rep < 10; ++rep) {
                CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_graph) best_graph = ms;
            }
            printf("shape %s, %3d dependent launches: stream %.2f us per launch, graph %.2f us per launch\n", shape ? "256 x 1024 threads + 40 KB LDS" : "1 x 64 threads", n, best_stream * 1e3 / n, best_graph * 1e3 / n);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
