#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int lane = threadIdx.x;
    int v = lane * 10;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xF, 0xF, false);  // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xF, 0xF, false);  // wave_shr:1
    int c = __builtin_amdgcn_update_dpp(-1, v, 0x101, 0xF, 0xF, false);  // row_shl:1
    out[lane] = a; out[64 + lane] = b; out[128+lane] = c;
}
int main() {
    int* d; hipMalloc(&d, 192 * 4); k<<<1, 64>>>(d); int h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int j = 0; j < 3; ++j) { for (int i = 0; i < 64; ++i) printf("%d ", h[j*64+i]); printf("\n"); }
    return 0;
}
