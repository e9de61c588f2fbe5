// How long does the host take to notice that a small D2H read-back has landed?  hipStreamSynchronize vs spinning on hipEventQuery vs spinning on the pinned word itself.
// hipcc --offload-arch=gfx950 -O2 tools/probes/sync_latency.hip -o build/sync_latency && build/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_work(uint32_t* p, int spin) { if (threadIdx.x == 0) { uint32_t v = p[0]; for (int i = 0; i < spin; ++i) v = v * 1664525u + 1013904223u; p[0] = v | 1u; } }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreate(&s);
    uint32_t* d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
    volatile uint32_t* h; hipHostMalloc((void**)&h, 1024, hipHostMallocDefault);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int reps = 2000, launches = 8, spin = 2000;       // 8 dependent launches of a few microseconds each, then the read-back: the shape of a small build
    for (int mode = 0; mode < 4; ++mode) {
        for (int w = 0; w < 50; ++w) { hipLaunchKernelGGL(k_work, 1, 64, 0, s, d, spin); hipStreamSynchronize(s); }
        const double t0 = now();
        for (int r = 0; r < reps; ++r) {
            for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_work, 1, 64, 0, s, d, spin);
            if (mode == 0) { hipMemcpyAsync((void*)h, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
            else if (mode == 1) { hipMemcpyAsync((void*)h, d, 4, hipMemcpyDeviceToHost, s); hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) { } }
            else if (mode == 2) { h[0] = 0u; hipMemcpyAsync((void*)h, d, 4, hipMemcpyDeviceToHost, s); while (h[0] == 0u) { } }
            else { hipStreamSynchronize(s); }                // no copy at all: the floor
        }
        hipStreamSynchronize(s);
        const double t1 = now();
        const char* names[4] = { "memcpyAsync + hipStreamSynchronize", "memcpyAsync + event + hipEventQuery spin", "memcpyAsync + spin on the pinned word", "hipStreamSynchronize only (no copy)" };
        printf("%-44s %8.2f us per iteration\n", names[mode], (t1 - t0) / reps);
    }
    return 0;
}
