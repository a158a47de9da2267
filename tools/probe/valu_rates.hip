// VALU issue cost per instruction class on gfx950 (round 5): each kernel runs a long unrolled stream of ONE instruction kind on independent registers, W waves per SIMD, and reports
// shader cycles per wave-instruction per SIMD = elapsed_cycles * waves_per_simd ... (see main).   hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body)                                                                            \
    __global__ __launch_bounds__(256) void name(float* out, int iters) {                               \
        float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        float b0 = 1.0001f, b1 = 0.9999f;                                                               \
        unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7; \
        for (int i = 0; i < iters; ++i) {                                                               \
            asm volatile(REP8(body)                                                                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) \
                         : "v"(b0), "v"(b1)                                                             \
                         : "vcc");                                                                      \
        }                                                                                               \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f || (u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) == 0x12345u) out[threadIdx.x] = a0;  \
    }
// operands: %0..%7 floats, %8..%15 uints, %16 %17 constants.  Each body = 8 independent instructions.
KERNEL(k_fma, "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n")
KERNEL(k_mul, "v_mul_f32 %0, %0, %16\n v_mul_f32 %1, %1, %16\n v_mul_f32 %2, %2, %16\n v_mul_f32 %3, %3, %16\n v_mul_f32 %4, %4, %16\n v_mul_f32 %5, %5, %16\n v_mul_f32 %6, %6, %16\n v_mul_f32 %7, %7, %16\n")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
KERNEL(k_mullo, "v_mul_lo_u32 %8, %8, %9\n v_mul_lo_u32 %9, %9, %10\n v_mul_lo_u32 %10, %10, %11\n v_mul_lo_u32 %11, %11, %12\n v_mul_lo_u32 %12, %12, %13\n v_mul_lo_u32 %13, %13, %14\n v_mul_lo_u32 %14, %14, %15\n v_mul_lo_u32 %15, %15, %8\n")
KERNEL(k_mul24, "v_mul_u32_u24 %8, %8, %9\n v_mul_u32_u24 %9, %9, %10\n v_mul_u32_u24 %10, %10, %11\n v_mul_u32_u24 %11, %11, %12\n v_mul_u32_u24 %12, %12, %13\n v_mul_u32_u24 %13, %13, %14\n v_mul_u32_u24 %14, %14, %15\n v_mul_u32_u24 %15, %15, %8\n")
KERNEL(k_mad24, "v_mad_u32_u24 %8, %8, %9, %10\n v_mad_u32_u24 %9, %9, %10, %11\n v_mad_u32_u24 %10, %10, %11, %12\n v_mad_u32_u24 %11, %11, %12, %13\n v_mad_u32_u24 %12, %12, %13, %14\n v_mad_u32_u24 %13, %13, %14, %15\n v_mad_u32_u24 %14, %14, %15, %8\n v_mad_u32_u24 %15, %15, %8, %9\n")
KERNEL(k_xor, "v_xor_b32 %8, %8, %9\n v_xor_b32 %9, %9, %10\n v_xor_b32 %10, %10, %11\n v_xor_b32 %11, %11, %12\n v_xor_b32 %12, %12, %13\n v_xor_b32 %13, %13, %14\n v_xor_b32 %14, %14, %15\n v_xor_b32 %15, %15, %8\n")
KERNEL(k_xorsdwa, "v_xor_b32_sdwa %8, %8, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %9, %9, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %10, %10, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %11, %11, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %12, %12, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %13, %13, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %14, %14, %14 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %15, %15, %15 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n")
KERNEL(k_bfi, "v_bfi_b32 %8, %9, %8, %10\n v_bfi_b32 %9, %10, %9, %11\n v_bfi_b32 %10, %11, %10, %12\n v_bfi_b32 %11, %12, %11, %13\n v_bfi_b32 %12, %13, %12, %14\n v_bfi_b32 %13, %14, %13, %15\n v_bfi_b32 %14, %15, %14, %8\n v_bfi_b32 %15, %8, %15, %9\n")
KERNEL(k_alignbit, "v_alignbit_b32 %8, %8, %8, 8\n v_alignbit_b32 %9, %9, %9, 8\n v_alignbit_b32 %10, %10, %10, 8\n v_alignbit_b32 %11, %11, %11, 8\n v_alignbit_b32 %12, %12, %12, 8\n v_alignbit_b32 %13, %13, %13, 8\n v_alignbit_b32 %14, %14, %14, 8\n v_alignbit_b32 %15, %15, %15, 8\n")
KERNEL(k_cvtpk, "v_cvt_pk_bf16_f32 %8, %0, %1\n v_cvt_pk_bf16_f32 %9, %1, %2\n v_cvt_pk_bf16_f32 %10, %2, %3\n v_cvt_pk_bf16_f32 %11, %3, %4\n v_cvt_pk_bf16_f32 %12, %4, %5\n v_cvt_pk_bf16_f32 %13, %5, %6\n v_cvt_pk_bf16_f32 %14, %6, %7\n v_cvt_pk_bf16_f32 %15, %7, %0\n")
KERNEL(k_cmp_cnd, "v_cmp_ge_u32 vcc, %8, %9\n v_cndmask_b32 %0, 0, %0, vcc\n v_cmp_ge_u32 vcc, %9, %10\n v_cndmask_b32 %1, 0, %1, vcc\n v_cmp_ge_u32 vcc, %10, %11\n v_cndmask_b32 %2, 0, %2, vcc\n v_cmp_ge_u32 vcc, %11, %12\n v_cndmask_b32 %3, 0, %3, vcc\n")
KERNEL(k_cmpsdwa_cnd, "v_cmp_ge_u32_sdwa vcc, %8, %9 src0_sel:WORD_1 src1_sel:DWORD\n v_cndmask_b32 %0, 0, %0, vcc\n v_cmp_ge_u32_sdwa vcc, %9, %10 src0_sel:WORD_0 src1_sel:DWORD\n v_cndmask_b32 %1, 0, %1, vcc\n v_cmp_ge_u32_sdwa vcc, %10, %11 src0_sel:WORD_1 src1_sel:DWORD\n v_cndmask_b32 %2, 0, %2, vcc\n v_cmp_ge_u32_sdwa vcc, %11, %12 src0_sel:WORD_0 src1_sel:DWORD\n v_cndmask_b32 %3, 0, %3, vcc\n")
KERNEL(k_permlane, "v_permlane32_swap_b32 %8, %9\n v_permlane32_swap_b32 %10, %11\n v_permlane32_swap_b32 %12, %13\n v_permlane32_swap_b32 %14, %15\n v_permlane32_swap_b32 %8, %10\n v_permlane32_swap_b32 %9, %11\n v_permlane32_swap_b32 %12, %14\n v_permlane32_swap_b32 %13, %15\n")
KERNEL(k_movdpp, "v_mov_b32_dpp %8, %9 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %9, %10 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %10, %11 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %11, %12 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %12, %13 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %13, %14 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %14, %15 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %15, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n")

// packed f32: register pairs
__global__ __launch_bounds__(256) void k_pkfma(float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {threadIdx.x * 1e-3f + 1.f, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b0 = {1.0001f, 0.9999f}, b1 = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 12345.f) out[threadIdx.x] = s.x;
}
__global__ __launch_bounds__(256) void k_pkmul(float* out, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {threadIdx.x * 1e-3f + 1.f, 2.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b0 = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 12345.f) out[threadIdx.x] = s.x;
}

typedef void (*kfn)(float*, int);
int main() {
    struct { const char* name; kfn f; } ks[] = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul}, {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp},
        {"v_mul_lo_u32", k_mullo}, {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_xor_b32", k_xor}, {"v_xor_b32_sdwa", k_xorsdwa}, {"v_bfi_b32", k_bfi}, {"v_alignbit_b32", k_alignbit},
        {"v_cvt_pk_bf16_f32", k_cvtpk}, {"v_cmp+v_cndmask (pair)", k_cmp_cnd}, {"v_cmp_sdwa+v_cndmask (pair)", k_cmpsdwa_cnd}, {"v_permlane32_swap", k_permlane}, {"v_mov_b32_dpp", k_movdpp}};
    float* out; hipMalloc(&out, 4096);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz nominal\n", prop.name, cus, ghz);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4}) {             // waves per SIMD: blocks of 256 threads (4 waves, one per SIMD) x wps blocks per CU
        printf("--- %d wave(s) per SIMD\n", wps);
        for (auto& k : ks) {
            k.f<<<cus * wps, 256>>>(out, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k.f<<<cus * wps, 256>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)iters * 64.0 * wps;          // wave-instructions per SIMD
            printf("  %-30s %8.1f us   %6.2f ns per wave-instruction per SIMD  = %5.2f cycles at %.2f GHz\n", k.name, ms * 1e3, ms * 1e6 / n, ms * 1e6 / n * ghz, ghz);
        }
    }
    return 0;
}
