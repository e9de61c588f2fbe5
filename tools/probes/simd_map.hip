// simd_map.hip — which SIMD does wave w of a 256-thread workgroup run on?  (probe for the HPB_ROT switch of csrc/hploc.hip)
// hipcc --offload-arch=gfx950 -O2 tools/probes/simd_map.hip -o /tmp/simd_map && /tmp/simd_map
// Launches workgroups shaped like k_hploc_block's (256 threads, 22.9 KB of LDS, 7 per CU) and prints, per wave index, the histogram of HW_ID.SIMD_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 7) void k(unsigned* out, int spin) {
    __shared__ unsigned pad[5860];
    if (threadIdx.x == 0 && spin < 0) pad[blockIdx.x % 5860] = 1;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID, all 32 bits
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);          // keep the workgroup resident for a while so that the CU fills up
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
}
int main() {
    const int G = 4096;
    unsigned* d; hipMalloc(&d, G * 4 * sizeof(unsigned));
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, 2000);
    std::vector<unsigned> h(G * 4); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    for (int b = 0; b < G; ++b) for (int w = 0; w < 4; ++w) hist[w][(h[b * 4 + w] >> 4) & 3]++;
    for (int w = 0; w < 4; ++w) printf("wave %d: SIMD0 %d  SIMD1 %d  SIMD2 %d  SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    // same SIMD for wave 0 of consecutive workgroups on one CU?
    int same = 0, tot = 0;
    for (int b = 0; b < G; ++b) { int s[4]; for (int w = 0; w < 4; ++w) s[w] = (h[b * 4 + w] >> 4) & 3; tot++; if (s[0] != s[1] && s[1] != s[2] && s[2] != s[3] && s[0] != s[2] && s[0] != s[3] && s[1] != s[3]) same++; }
    printf("workgroups whose four waves sit on four different SIMDs: %d of %d\n", same, tot);
    return 0;
}
