// Do VALU instructions overlap v_mfma_f32_32x32x16_bf16 on ONE SIMD of gfx950 -- across waves, and inside a wave?  (round 5; the fused MLP kernels' MFMA and VALU times ADD.)
//   hipcc --offload-arch=gfx950 -O2 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// mode bits per wave:  1 = MFMA stream (two independent accumulator chains), 2 = VALU stream (8 independent v_fma chains), 3 = both interleaved in the wave (1 MFMA : NV fma)
template <int NV, bool CHAIN>
__device__ __forceinline__ void mfma_body(f16v& c0, f16v& c1, bf8 a, bf8 b, float (&v)[8], float k0, float k1) {
    // one MFMA (alternating accumulators unless CHAIN) followed by NV fmas
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (CHAIN || (i & 1) == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        else c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * NV + j) & 7]) : "v"(k0), "v"(k1));
    }
}

template <int NV, bool CHAIN>
__global__ __launch_bounds__(512) void k_mix(float* out, int iters, int mfma_waves_mask, int valu_waves_mask) {
    const int wave = threadIdx.x >> 6;
    f16v c0, c1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    bf8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x & 7)); b[i] = (__bf16)0.5f; }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 1.f + i + threadIdx.x * 1e-3f;
    const float k0 = 0.9999f, k1 = 0.0001f;
    const bool do_m = (mfma_waves_mask >> wave) & 1, do_v = (valu_waves_mask >> wave) & 1;      // wave-uniform
    if (do_m && do_v) {
        for (int it = 0; it < iters; ++it) mfma_body<NV, CHAIN>(c0, c1, a, b, v, k0, k1);
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) mfma_body<0, CHAIN>(c0, c1, a, b, v, k0, k1);
    } else if (do_v) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16 * (NV > 0 ? NV : 8); ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(k0), "v"(k1));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int NV, bool CHAIN>
static float run(float* out, int blocks, int threads, int iters, int mm, int vm) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_mix<NV, CHAIN><<<blocks, threads>>>(out, 4, mm, vm);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_mix<NV, CHAIN><<<blocks, threads>>>(out, iters, mm, vm);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 2000;
    printf("%d CUs; per wave and launch: %d x 16 MFMAs (32x32x16 bf16), VALU streams: 16 x NV v_fma_f32 per 16 MFMAs\n", cus, iters);
    // A. across waves, 512-thread blocks = 2 waves per SIMD (waves w and w + 4 share SIMD w under the usual placement)
    printf("A. two waves per SIMD (one 512-thread block per CU), NV = 8 fma per MFMA-equivalent\n");
    printf("   MFMA in waves 0-3 only (independent pairs)      %8.1f us\n", run<8, false>(out, cus, 512, iters, 0x0f, 0x00));
    printf("   MFMA in waves 0-3 only (one dependent chain)    %8.1f us\n", run<8, true>(out, cus, 512, iters, 0x0f, 0x00));
    printf("   VALU in waves 4-7 only                          %8.1f us\n", run<8, false>(out, cus, 512, iters, 0x00, 0xf0));
    printf("   MFMA in waves 0-3 + VALU in waves 4-7           %8.1f us\n", run<8, false>(out, cus, 512, iters, 0x0f, 0xf0));
    printf("   MFMA (chain) in waves 0-3 + VALU in waves 4-7   %8.1f us\n", run<8, true>(out, cus, 512, iters, 0x0f, 0xf0));
    printf("   MFMA in all 8 waves                             %8.1f us\n", run<8, false>(out, cus, 512, iters, 0xff, 0x00));
    printf("   VALU in all 8 waves                             %8.1f us\n", run<8, false>(out, cus, 512, iters, 0x00, 0xff));
    // B. inside a wave: 1 MFMA : NV fma
    printf("B. in-wave interleave, one 256-thread block per CU (1 wave per SIMD)\n");
    printf("   NV = 0  %8.1f us\n", run<0, false>(out, cus, 256, iters, 0x0f, 0x0f));
    printf("   NV = 2  %8.1f us\n", run<2, false>(out, cus, 256, iters, 0x0f, 0x0f));
    printf("   NV = 4  %8.1f us\n", run<4, false>(out, cus, 256, iters, 0x0f, 0x0f));
    printf("   NV = 6  %8.1f us\n", run<6, false>(out, cus, 256, iters, 0x0f, 0x0f));
    printf("   NV = 8  %8.1f us\n", run<8, false>(out, cus, 256, iters, 0x0f, 0x0f));
    printf("   NV = 12 %8.1f us\n", run<12, false>(out, cus, 256, iters, 0x0f, 0x0f));
    printf("   NV = 16 %8.1f us\n", run<16, false>(out, cus, 256, iters, 0x0f, 0x0f));
    printf("   NV = 16, dependent chain %8.1f us\n", run<16, true>(out, cus, 256, iters, 0x0f, 0x0f));
    printf("   VALU alone, 16 x 16 fma per iteration %8.1f us\n", run<16, false>(out, cus, 256, iters, 0x00, 0x0f));
    printf("C. in-wave interleave at 2 waves per SIMD (512-thread blocks), all 8 waves run both\n");
    printf("   NV = 8  %8.1f us\n", run<8, false>(out, cus, 512, iters, 0xff, 0xff));
    printf("   NV = 16 %8.1f us\n", run<16, false>(out, cus, 512, iters, 0xff, 0xff));
    printf("   NV = 16, dependent chain %8.1f us\n", run<16, true>(out, cus, 512, iters, 0xff, 0xff));
    printf("   VALU alone NV = 16 %8.1f us;  MFMA alone %8.1f us\n", run<16, false>(out, cus, 512, iters, 0x00, 0xff), run<16, false>(out, cus, 512, iters, 0xff, 0x00));
    return 0;
}
