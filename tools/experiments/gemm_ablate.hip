// Ablation of the plane NT GEMM main loop (gemm_bp.hip structure: 128x128 tile, 4 waves, BK = 32, two LDS stages, one barrier per
// slab, global_load_lds staging, bf16x3 = 3 MFMAs per product).  Each variant removes one ingredient; values are kept alive with
// empty asm so the compiler cannot delete the rest.   hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_ablate.hip -o gemm_ablate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

enum { FULL = 0, NO_GLDS = 1, NO_DSREAD = 2, MFMA_ONLY = 3, NO_MFMA = 4, NO_BARRIER_NO_GLDS = 5, ONE_MFMA = 6, FULL_SYNCTHREADS = 7 };

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int ABL, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ C,
                                                                                 int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int BM = 128, BN = 128, P = 2;
    constexpr int A_BYTES = P * BM * 64, STAGE = A_BYTES + P * BN * 64;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tn = tile % tiles_n, tm = tile / tiles_n, m0 = tm * BM, n0 = tn * BN;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    const int prow = lane >> 2, pchunk = (lane & 3) ^ ((lane >> 4) & 3);
    const long a_plane = (long)M * K, b_plane = (long)N * K;
    auto issue = [&](int k0, int buf) __attribute__((always_inline)) {
        if (ABL == NO_GLDS || ABL == MFMA_ONLY || ABL == NO_BARRIER_NO_GLDS) return;
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int q0 = 0; q0 < 4; ++q0) {
            const int q = q0 * 4 + wave, pl = q / 8, rq = q % 8;
            const uint16_t* g = A + pl * a_plane + (long)(m0 + rq * 16 + prow) * K + k0 + pchunk * 8;
            __builtin_amdgcn_global_load_lds(g, base + pl * BM * 64 + rq * 1024, 16, 0, 0);
        }
#pragma unroll
        for (int q0 = 0; q0 < 4; ++q0) {
            const int q = q0 * 4 + wave, pl = q / 8, rq = q % 8;
            const uint16_t* g = B + pl * b_plane + (long)(n0 + rq * 16 + prow) * K + k0 + pchunk * 8;
            __builtin_amdgcn_global_load_lds(g, base + A_BYTES + pl * BN * 64 + rq * 1024, 16, 0, 0);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][2], fb[2][2];      // constant fragments for the NO_DSREAD variants
    {
        u4 v = {(unsigned)tid * 2654435761u, (unsigned)lane, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) { fa[i][pl] = __builtin_bit_cast(bf16x8, v); fb[i][pl] = __builtin_bit_cast(bf16x8, v); }
    }
    issue(0, 0);
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += 32) {
        if (ABL != NO_BARRIER_NO_GLDS) {
            if (ABL == FULL_SYNCTHREADS) __syncthreads();
            else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        }
        if (k0 + 32 < K) issue(k0 + 32, buf ^ 1);
        const char* As_ = smem + buf * STAGE; const char* Bs_ = As_ + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = 2 * ks + lhi;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (ABL == NO_DSREAD || ABL == MFMA_ONLY) { ah[i] = fa[i][0]; al[i] = fa[i][1]; asm volatile("" : "+v"(ah[i]), "+v"(al[i])); continue; }
                const int r = wm0 + i * 32 + l31, off = r * 64 + ((c ^ ((r >> 2) & 3)) * 16);
                ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(As_ + off));
                al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(As_ + BM * 64 + off));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (ABL == NO_DSREAD || ABL == MFMA_ONLY) { bh[j] = fb[j][0]; bl[j] = fb[j][1]; asm volatile("" : "+v"(bh[j]), "+v"(bl[j])); continue; }
                const int r = wn0 + j * 32 + l31, off = r * 64 + ((c ^ ((r >> 2) & 3)) * 16);
                bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(Bs_ + off));
                bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u4*>(Bs_ + BN * 64 + off));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (ABL == NO_MFMA) { asm volatile("" :: "v"(ah[i]), "v"(al[i]), "v"(bh[j]), "v"(bl[j])); continue; }
                    if (ABL != ONE_MFMA) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    } else asm volatile("" :: "v"(al[i]), "v"(bl[j]));
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                }
        }
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + wm0 + i * 32 + l31, col = n0 + wn0 + j * 32 + 8 * q + 4 * lhi;
                *reinterpret_cast<float4*>(C + (long)row * N + col) = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            }
}

__global__ void fill(uint16_t* p, long n, unsigned seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
        p[i] = (uint16_t)(0x3c00 + (x & 0x03ff) + ((x >> 10 & 1) << 15));          // +-[0.0078, 0.0156): random sign and mantissa
    }
}

template <int ABL, int WPE>
float run(const uint16_t* A, const uint16_t* B, float* C, int M, int N, int K) {
    const int tm = M / 128, tn = N / 128;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<ABL, WPE>), dim3(tm * tn), dim3(256), 0, 0, A, B, C, M, N, K, tm, tn);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<ABL, WPE>), dim3(tm * tn), dim3(256), 0, 0, A, B, C, M, N, K, tm, tn);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / it * 1e3f;
}

int main() {
    const int shapes[][3] = {{16384, 1280, 320}, {32768, 1280, 320}, {32768, 1280, 1280}, {8192, 2048, 512}, {65536, 1024, 128}, {16384, 1280, 2560}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        uint16_t *A, *B; float* C;
        hipMalloc(&A, 2L * 2 * M * K); hipMalloc(&B, 2L * 2 * N * K); hipMalloc(&C, 4L * M * N);
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, A, 2L * M * K, 1u);
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B, 2L * N * K, 2u);
        const double fl = 2.0 * M * N * K;
        printf("M=%d N=%d K=%d  (useful TF/s; bf16x3 roof 833, per-MFMA roof 2500)\n", M, N, K);
#define R(ABL, WPE, name) { float us = run<ABL, WPE>(A, B, C, M, N, K); printf("  %-34s %8.1f us  %6.1f TF/s\n", name, us, fl / us / 1e6); }
        R(FULL, 2, "full (2 waves/SIMD)");
        R(FULL_SYNCTHREADS, 2, "full, __syncthreads");
        R(NO_GLDS, 2, "no global loads");
        R(NO_DSREAD, 2, "no ds_read (glds + mfma)");
        R(MFMA_ONLY, 2, "mfma + barrier only");
        R(NO_BARRIER_NO_GLDS, 2, "ds_read + mfma, no barrier/glds");
        R(NO_MFMA, 2, "no mfma (glds + ds_read)");
        R(ONE_MFMA, 2, "1 mfma per product (bf16)");
        hipFree(A); hipFree(B); hipFree(C);
    }
    printf("status %s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
