// How many 256-thread workgroups does a gfx950 CU really hold as a function of their LDS bytes (is an 8th HPLOC tile per CU possible at ~20 KB)?
// API answer (hipOccupancyMaxActiveBlocksPerMultiprocessor) and measured answer: 8192 workgroups that each raise a global counter, spin ~30 us and
// lower it again; peak / 256 CUs = resident workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 8) void k(int* c) {
    extern __shared__ int s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int now = atomicAdd(c, 1) + 1;
        atomicMax(c + 1, now);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 3000ull) __builtin_amdgcn_s_sleep(8);     // 100 MHz clock: 30 us
        atomicAdd(c, -1 + (s[255] == 12345 ? 1 : 0));
    }
}
int main() {
    int* d; (void)hipMalloc(&d, 8);
    for (int b = 17408; b <= 23552; b += 256) {
        int nb = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, b);
        (void)hipMemset(d, 0, 8);
        hipLaunchKernelGGL(k, dim3(8192), dim3(256), b, 0, d);
        int h[2]; (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("%5d B of LDS: API %d workgroups per CU, measured peak %d = %.2f per CU\n", b, nb, h[1], h[1] / 256.0);
    }
    return 0;
}
