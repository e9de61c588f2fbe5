// ubench_issue.hip — issue cost (SIMD cycles per wave64 instruction) of the instruction kinds the emit kernels are made of, on gfx950,
// at 1 / 2 / 4 / 7 waves per SIMD.  Measurement tool, not part of the library.  Build + run (GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o build/ubench_issue && build/ubench_issue
// Each kernel runs ITERS x 32 instructions of one kind on 8 independent destination registers; the wave's elapsed s_memtime (shader cycles)
// is recorded; cycles per instruction per SIMD = max elapsed / (instructions per wave x waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>

#define R8(op) op(10) op(11) op(12) op(13) op(14) op(15) op(16) op(17)
#define R32(op) R8(op) R8(op) R8(op) R8(op)
#define P8(op) op(10) op(12) op(14) op(16) op(18) op(20) op(22) op(24)
#define P32(op) P8(op) P8(op) P8(op) P8(op)
#define Q8(op) op(12) op(16) op(20) op(24) op(12) op(16) op(20) op(24)
#define Q32(op) Q8(op) Q8(op) Q8(op) Q8(op)
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "vcc", "scc", "s20", "s21", "s22", "s23", "memory"

#define I_MIN(d)      "v_min_f32 v" #d ", v1, v" #d "\n"
#define I_MINDPP(d)   "v_min_f32_dpp v" #d ", v1, v" #d " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MAXDPPR(d)  "v_max_f32_dpp v" #d ", v1, v" #d " row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MOVWSHL(d)  "v_mov_b32_dpp v" #d ", v1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MOVRSHL(d)  "v_mov_b32_dpp v" #d ", v1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_MOV(d)      "v_mov_b32 v" #d ", v1\n"
#define I_PKMUL(d)    "v_pk_mul_f32 v[" #d ":" #d "+1], v[4:5], v[6:7]\n"
#define I_PKADD(d)    "v_pk_add_f32 v[" #d ":" #d "+1], v[4:5], v[6:7]\n"
#define I_FMA(d)      "v_fma_f32 v" #d ", v1, v3, v" #d "\n"
#define I_MUL(d)      "v_mul_f32 v" #d ", v1, v" #d "\n"
#define I_CNDMASK(d)  "v_cndmask_b32 v" #d ", v1, v3, vcc\n"
#define I_CMPCND(d)   "v_cmp_lt_u32 vcc, v1, v" #d "\nv_cndmask_b32 v" #d ", v1, v3, vcc\n"
#define I_CMPS(d)     "v_cmp_lt_u32_e64 s[20:21], v1, v" #d "\n"
#define I_BPERM(d)    "ds_bpermute_b32 v" #d ", v2, v1\n"
#define I_PERM(d)     "ds_permute_b32 v" #d ", v2, v1\n"
#define I_DSMIN64(d)  "ds_min_u64 v2, v[4:5]\n"
#define I_DSMINR64(d) "ds_min_rtn_u64 v[" #d ":" #d "+1], v2, v[4:5]\n"
#define I_DSRD(d)     "ds_read_b32 v" #d ", v2\n"
#define I_DSRD64(d)   "ds_read_b64 v[" #d ":" #d "+1], v2\n"
#define I_DSWR(d)     "ds_write_b32 v2, v1\n"
#define I_DSWR64(d)   "ds_write_b64 v2, v[4:5]\n"
#define I_MIN3(d)     "v_min3_f32 v" #d ", v1, v3, v" #d "\n"
#define I_PL32(d)     "v_permlane32_swap_b32 v" #d ", v1\n"
#define I_PL16(d)     "v_permlane16_swap_b32 v" #d ", v1\n"
#define I_RDLANE(d)   "v_readlane_b32 s20, v" #d ", 5\n"
#define I_ADDU(d)     "v_add_u32 v" #d ", v1, v" #d "\n"
#define I_MINU(d)     "v_min_u32 v" #d ", v1, v" #d "\n"
#define I_MINUDPP(d)  "v_min_u32_dpp v" #d ", v1, v" #d " row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_BFE(d)      "v_bfe_u32 v" #d ", v1, 3, 5\n"
#define I_LSHLOR(d)   "v_lshl_or_b32 v" #d ", v1, 3, v3\n"
#define I_MIX1(d)     "v_min_f32 v" #d ", v1, v" #d "\nds_bpermute_b32 v18, v2, v1\n"
#define I_MIX2(d)     "v_min_f32 v" #d ", v1, v" #d "\ns_add_u32 s22, s22, 1\n"
#define I_MIX3(d)     "v_min_f32 v" #d ", v1, v" #d "\nv_min_f32 v18, v1, v3\nv_min_f32 v19, v1, v3\nds_min_u64 v2, v[4:5]\n"
#define I_CMPEQDPP(d) "v_cmp_eq_u32_dpp vcc, v1, v" #d " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_SALU(d)     "s_add_u32 s22, s22, 1\n"
#define I_MBCNT(d)    "v_mbcnt_lo_u32_b32 v" #d ", s22, 0\n"
#define I_MAX3(d)     "v_max3_f32 v" #d ", v1, v3, v" #d "\n"
#define I_SUBMIN(d)   "v_sub_f32 v" #d ", v1, v" #d "\n"
#define I_PKFMA(d)    "v_pk_fma_f32 v[" #d ":" #d "+1], v[4:5], v[6:7], v[" #d ":" #d "+1]\n"
#define I_SWZ(d)      "ds_swizzle_b32 v" #d ", v1 offset:swizzle(SWAP,16)\n"
#define I_CNDS(d)     "v_cndmask_b32_e64 v" #d ", v1, v3, s[20:21]\n"
#define I_CNDS2(d)    "v_cndmask_b32_e64 v" #d ", v1, v3, s[20:21]\nv_cndmask_b32_e64 v18, v1, v3, s[22:23]\n"
#define I_SUBF(d)     "v_sub_f32 v" #d ", v1, v" #d "\n"
#define I_ADDF(d)     "v_add_f32 v" #d ", v1, v" #d "\n"
#define I_CMPV(d)     "v_cmp_lt_u32 vcc, v1, v" #d "\n"
#define I_MINI(d)     "v_min_i32 v" #d ", v1, v" #d "\n"
#define I_MAXF(d)     "v_max_f32 v" #d ", v1, v" #d "\n"
#define I_MED3(d)     "v_med3_f32 v" #d ", v1, v3, v" #d "\n"
#define I_DSRD128(d)  "ds_read_b128 v[" #d ":" #d "+3], v8\n"
#define I_DSWR128(d)  "ds_write_b128 v8, v[4:7]\n"
#define I_DSMIN32(d)  "ds_min_u32 v2, v1\n"
#define I_DSMINR32(d) "ds_min_rtn_u32 v" #d ", v2, v1\n"
#define I_DSADD32(d)  "ds_add_u32 v2, v1\n"
#define I_ANDB(d)     "v_and_b32 v" #d ", v1, v" #d "\n"
#define I_LSHL(d)     "v_lshlrev_b32 v" #d ", 3, v" #d "\n"
#define I_BCNT(d)     "v_bcnt_u32_b32 v" #d ", v1, v" #d "\n"
#define I_MINDPP0(d)  "v_min_f32_dpp v" #d ", v1, v" #d " row_shl:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDCO(d)    "v_addc_co_u32 v" #d ", vcc, v1, v" #d ", vcc\n"
#define I_MOV64(d)    "v_mov_b64 v[" #d ":" #d "+1], v[4:5]\n"

constexpr int ITERS = 2000;

#define KERNEL(NAME, BODY, PER, LDSOP)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out) {                                               \
        __shared__ unsigned long long s_pad[2048 + 64];                                                                        \
        s_pad[threadIdx.x] = 0; s_pad[threadIdx.x + 256] = 0; __syncthreads();                                            \
        asm volatile("v_mov_b32 v1, 1.5\nv_mov_b32 v3, 2.5\nv_mov_b32 v4, 1.0\nv_mov_b32 v5, 1.0\nv_mov_b32 v6, 1.0\nv_mov_b32 v7, 1.0\n" \
                     "v_mbcnt_lo_u32_b32 v2, -1, 0\nv_mbcnt_hi_u32_b32 v2, -1, v2\nv_lshlrev_b32 v2, 3, v2\n"             \
                     "v_lshl_add_u32 v2, %0, 9, v2\ns_mov_b32 s22, 0\nv_lshlrev_b32 v8, 1, v2\ns_mov_b64 s[20:21], 0x5555\n" :: "v"((unsigned)(threadIdx.x >> 6)) : "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "s20", "s21", "s22"); \
        unsigned long long t0, t1;                                                                                        \
        asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");                                        \
        for (int i = 0; i < ITERS; ++i) {                                                                                 \
            asm volatile(BODY ::: "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", CLOB);                                        \
            if (LDSOP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
        }                                                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                  \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                  \
        if (t1 == 12345) out[0] = s_pad[threadIdx.x];                                                                     \
    }

KERNEL(k_min, R32(I_MIN), 32, 0)
KERNEL(k_mindpp, R32(I_MINDPP), 32, 0)
KERNEL(k_maxdppr, R32(I_MAXDPPR), 32, 0)
KERNEL(k_movwshl, R32(I_MOVWSHL), 32, 0)
KERNEL(k_movrshl, R32(I_MOVRSHL), 32, 0)
KERNEL(k_mov, R32(I_MOV), 32, 0)
KERNEL(k_pkmul, P32(I_PKMUL), 32, 0)
KERNEL(k_pkadd, P32(I_PKADD), 32, 0)
KERNEL(k_pkfma, P32(I_PKFMA), 32, 0)
KERNEL(k_fma, R32(I_FMA), 32, 0)
KERNEL(k_mul, R32(I_MUL), 32, 0)
KERNEL(k_cndmask, R32(I_CNDMASK), 32, 0)
KERNEL(k_cmpcnd, R32(I_CMPCND), 64, 0)
KERNEL(k_cmps, R32(I_CMPS), 32, 0)
KERNEL(k_bperm, R32(I_BPERM), 32, 1)
KERNEL(k_perm, R32(I_PERM), 32, 1)
KERNEL(k_dsmin64, R32(I_DSMIN64), 32, 1)
KERNEL(k_dsminr64, P32(I_DSMINR64), 32, 1)
KERNEL(k_dsrd, R32(I_DSRD), 32, 1)
KERNEL(k_dsrd64, P32(I_DSRD64), 32, 1)
KERNEL(k_dswr, R32(I_DSWR), 32, 1)
KERNEL(k_dswr64, R32(I_DSWR64), 32, 1)
KERNEL(k_min3, R32(I_MIN3), 32, 0)
KERNEL(k_max3, R32(I_MAX3), 32, 0)
KERNEL(k_pl32, R32(I_PL32), 32, 0)
KERNEL(k_pl16, R32(I_PL16), 32, 0)
KERNEL(k_rdlane, R32(I_RDLANE), 32, 0)
KERNEL(k_addu, R32(I_ADDU), 32, 0)
KERNEL(k_minu, R32(I_MINU), 32, 0)
KERNEL(k_minudpp, R32(I_MINUDPP), 32, 0)
KERNEL(k_bfe, R32(I_BFE), 32, 0)
KERNEL(k_lshlor, R32(I_LSHLOR), 32, 0)
KERNEL(k_mix_valu_bperm, R32(I_MIX1), 64, 1)
KERNEL(k_mix_valu_salu, R32(I_MIX2), 64, 0)
KERNEL(k_mix_3valu_dsmin, R32(I_MIX3), 128, 1)
KERNEL(k_salu, R32(I_SALU), 32, 0)
KERNEL(k_mbcnt, R32(I_MBCNT), 32, 0)
KERNEL(k_swizzle, R32(I_SWZ), 32, 1)
KERNEL(k_mov64, P32(I_MOV64), 32, 0)
KERNEL(k_cnd_sgpr, R32(I_CNDS), 32, 0)
KERNEL(k_cnd_2sgpr, R32(I_CNDS2), 64, 0)
KERNEL(k_subf, R32(I_SUBF), 32, 0)
KERNEL(k_addf, R32(I_ADDF), 32, 0)
KERNEL(k_cmpv, R32(I_CMPV), 32, 0)
KERNEL(k_mini, R32(I_MINI), 32, 0)
KERNEL(k_maxf, R32(I_MAXF), 32, 0)
KERNEL(k_med3, R32(I_MED3), 32, 0)
KERNEL(k_dsrd128, Q32(I_DSRD128), 32, 1)
KERNEL(k_dswr128, R32(I_DSWR128), 32, 1)
KERNEL(k_dsmin32, R32(I_DSMIN32), 32, 1)
KERNEL(k_dsminr32, R32(I_DSMINR32), 32, 1)
KERNEL(k_dsadd32, R32(I_DSADD32), 32, 1)
KERNEL(k_andb, R32(I_ANDB), 32, 0)
KERNEL(k_lshl, R32(I_LSHL), 32, 0)
KERNEL(k_bcnt, R32(I_BCNT), 32, 0)
KERNEL(k_mindpp_nobc, R32(I_MINDPP0), 32, 0)
KERNEL(k_addco, R32(I_ADDCO), 32, 0)

struct Entry { const char* name; void (*fn)(unsigned long long*); int per; };
#define E(NAME, PER) { #NAME, NAME, PER }
static Entry entries[] = {
    E(k_min, 32), E(k_mul, 32), E(k_fma, 32), E(k_mov, 32), E(k_mov64, 32), E(k_addu, 32), E(k_minu, 32), E(k_bfe, 32), E(k_lshlor, 32), E(k_min3, 32), E(k_max3, 32),
    E(k_mindpp, 32), E(k_mindpp_nobc, 32), E(k_maxdppr, 32), E(k_minudpp, 32), E(k_movwshl, 32), E(k_movrshl, 32),
    E(k_subf, 32), E(k_addf, 32), E(k_maxf, 32), E(k_mini, 32), E(k_med3, 32), E(k_andb, 32), E(k_lshl, 32), E(k_bcnt, 32), E(k_cmpv, 32), E(k_addco, 32), E(k_cnd_sgpr, 32), E(k_cnd_2sgpr, 64),
    E(k_dsrd128, 32), E(k_dswr128, 32), E(k_dsmin32, 32), E(k_dsminr32, 32), E(k_dsadd32, 32),
    E(k_pkmul, 32), E(k_pkadd, 32), E(k_pkfma, 32),
    E(k_cndmask, 32), E(k_cmpcnd, 64), E(k_cmps, 32), E(k_rdlane, 32), E(k_mbcnt, 32), E(k_salu, 32),
    E(k_pl32, 32), E(k_pl16, 32),
    E(k_bperm, 32), E(k_perm, 32), E(k_swizzle, 32), E(k_dsmin64, 32), E(k_dsminr64, 32), E(k_dsrd, 32), E(k_dsrd64, 32), E(k_dswr, 32), E(k_dswr64, 32),
    E(k_mix_valu_bperm, 64), E(k_mix_valu_salu, 64), E(k_mix_3valu_dsmin, 128),
};

int main() {
    unsigned long long* d; hipMalloc(&d, 8 * 256 * 8 * 4);
    std::vector<unsigned long long> h(256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("| kernel | instr/body | s_memtime ticks per instruction per SIMD at 1 / 2 / 4 / 8 waves per SIMD | wall ns per instruction per SIMD (x 2.4 = cycles at 2.4 GHz) at 1 / 2 / 4 / 8 waves per SIMD |\n|---|---|---|---|\n");
    for (auto& e : entries) {
        printf("| %s | %d |", e.name, e.per);
        double wall[4]; int wi = 0;
        for (int k : {1, 2, 4, 8}) {
            const int blocks = 256 * k;
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d);     // warm-up
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            wall[wi++] = (double)ms * 1e6 / ((double)ITERS * e.per * k);
            hipMemcpy(h.data(), d, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
            std::vector<unsigned long long> v(h.begin(), h.begin() + blocks * 4);
            std::sort(v.begin(), v.end());
            const double med = (double)v[v.size() / 2];
            const double per_simd = med / ((double)ITERS * e.per * k);
            printf(" %.2f", per_simd);
        }
        printf(" | %.3f %.3f %.3f %.3f |\n", wall[0], wall[1], wall[2], wall[3]); fflush(stdout);
    }
    hipFree(d);
    return 0;
}
