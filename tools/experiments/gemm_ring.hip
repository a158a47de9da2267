// "Ring" GEMMs: the same bf16x3 / bf16 arithmetic as gemm.hip / gemm_bp.hip, built around what measurement showed to be the limit
// of those kernels on this chip -- not VALU, not the matrix cores, but the number of bytes a CU has in flight from L2 / HBM
// (one K slab per workgroup: ~64 KB per CU, ~2 us per slab under load => ~8 TB/s of L2->CU traffic at 32 flop/byte tiles).
// Here every operand slab goes HBM/L2 -> LDS by global_load_lds into a ring of STAGES buffers with COUNTED s_waitcnt vmcnt and a
// raw s_barrier (one per slab): STAGES-1 slabs are in flight while one is multiplied, on tiles with more flops per loaded byte.
//   NT  C[M,N] = A[M,K] B[N,K]^T : A stays fp32 in HBM and in LDS (raw [row][32 k] image, 16-byte chunks XOR-swizzled by (row>>1)&7);
//       a lane reads its 8 k values as two ds_read_b128 and splits them hi/lo in registers (~20 VALU per fragment, next to
//       12 MFMAs).  B = pre-split weight planes (gemm_bp.hip's image).  256x128 tile, 8 waves, 3 x 48 KB stages.
//   TN  C[M,N] (+)= A[K,M]^T B[K,N] (weight gradients; both operands token-major fp32): raw [32 tokens][BM|BN] fp32 images, a
//       lane gathers the 8 tokens of its fragment with ds_read_b32 (lanes along m: conflict-free) and splits in registers.
//       128x128 tile, 4 waves, 4 x 32 KB stages; K split into slabs + fixed-order reduce as in gemm.hip; the bias gradient
//       (column sums of A) rides along from the LDS image.
// Replaces the same reference call sites as gemm.hip (nn.Linear / 1x1 conv forward and backward: mdvit.py:288,310-311,
// mpvit.py:71-78, Decoders.py:196,319-331).
#include "common.h"

typedef float rg_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 rg_bf16x8 __attribute__((ext_vector_type(8)));
// LDS fragment reads go through ext_vector types: HIP's float4 / uint4 are union-based structs whose accesses carry char-like
// TBAA, and hipcc then assumes every such ds_read may alias the in-flight global_load_lds writes -- it puts s_waitcnt vmcnt(0)
// in front of the first read of each slab, which drains the ring.  Element-typed vectors keep the counted waits below intact.
typedef float rg_f4 __attribute__((ext_vector_type(4)));
typedef unsigned rg_u4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;

struct RingArgs {
    const float* A; long lda;
    const void* B; long ldb; long b_plane;          // NT: bf16 planes [P][N][K];  TN: fp32 [K][N]
    int M, N, K;
    float* C; long ldc;
    uint16_t* Cp; long ldcp; long c_plane;
    float* U; long ldu_out;
    const float* bias;
    int e_drop; uint32_t e_k0, e_k1, e_thresh; float e_inv_keep;
    const float* e_rowscale; int e_rows_per_scale;
    const float* residual; long ldr;
    const float* gelu_u; long ldu;
    int splits; int k_per_split; float* slab;
    int accumulate;
    float* colsum;
    const uint32_t* seed;
    int tiles_m, tiles_n;
};

enum { REPI_PLAIN = 0, REPI_GELU = 1, REPI_DGELU = 2, REPI_FULL = 3 };

__device__ __forceinline__ int rg_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int N>
__device__ __forceinline__ void rg_wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else static_assert(N < 0, "add the vmcnt literal");
}

__device__ __forceinline__ void rg_split8(const float4 a, const float4 b, rg_bf16x8& hi, rg_bf16x8& lo) {
    uint2 h0, l0, h1, l1;
    mdvit_split_bf16x3(a, h0, l0);
    mdvit_split_bf16x3(b, h1, l1);
    hi = __builtin_bit_cast(rg_bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
    lo = __builtin_bit_cast(rg_bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
}

// ---------------------------------------------------------------------------------------------------------------------------
// NT
// ---------------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int P, int STAGES, int EPI>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64) void gemm_ring_nt_kernel(RingArgs p) {
    constexpr int WAVES_N = BN / 64, NW = (BM / 64) * WAVES_N;
    constexpr int A_BYTES = BM * 128, B_BYTES = P * BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BM / 8, NB = P * BN / 16;                        // 1 KiB pieces per stage
    static_assert(NA % NW == 0 && NB % NW == 0, "pieces must deal evenly to the waves (counted vmcnt)");
    constexpr int G = (NA + NB) / NW;                                    // global_load_lds instructions per wave per stage
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t ek0 = p.e_k0 ^ s0, ek1 = p.e_k1 + s1;
    const int tile = rg_xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg) / BK;
    const int wm0 = (wave / WAVES_N) * 64, wn0 = (wave % WAVES_N) * 64;
    const uint16_t* Bp = reinterpret_cast<const uint16_t*>(p.B);

    auto issue = [&](int ki, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE;
        const int k0 = kbeg + ki * BK;
#pragma unroll
        for (int q0 = 0; q0 < NA / NW; ++q0) {                          // A: piece = 8 rows x 128 B; lane i -> row i>>3, physical chunk i&7
            const int q = q0 * NW + wave;
            const int f = (((q & 1) << 2) + (lane >> 4)) & 7;           // (row >> 1) & 7 of row 8q + (lane >> 3)
            int row = m0 + q * 8 + (lane >> 3);
            row = row < p.M ? row : p.M - 1;
            const float* g = p.A + (long)row * p.lda + k0 + (((lane & 7) ^ f) << 2);
            __builtin_amdgcn_global_load_lds(g, base + q * 1024, 16, 0, 0);
        }
#pragma unroll
        for (int q0 = 0; q0 < NB / NW; ++q0) {                          // B planes: piece = 16 rows x 64 B (gemm_bp.hip's image)
            const int q = q0 * NW + wave;
            const int pl = q / (BN / 16), rq = q % (BN / 16);
            int row = n0 + rq * 16 + (lane >> 2);
            row = row < p.N ? row : p.N - 1;
            const uint16_t* g = Bp + pl * p.b_plane + (long)row * p.ldb + k0 + (((lane & 3) ^ ((lane >> 4) & 3)) << 3);
            __builtin_amdgcn_global_load_lds(g, base + A_BYTES + pl * BN * 64 + rq * 1024, 16, 0, 0);
        }
    };

    rg_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: STAGES-1 slabs in flight
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s, s);
    for (int ki = 0; ki < nk; ++ki) {
        // slab ki has landed once at most the younger slabs are outstanding: (slabs issued beyond ki) x G
        const int ahead = min(nk - 1 - ki, STAGES - 2);
        if (STAGES >= 4 && ahead >= 2) rg_wait_vmcnt<2 * G>();
        else if (ahead >= 1) rg_wait_vmcnt<G>();
        else rg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                  // everybody's pieces of slab ki are in; everybody is done with slab ki-1
        if (ki + STAGES - 1 < nk) issue(ki + STAGES - 1, (ki + STAGES - 1) % STAGES);
        const char* As_ = smem + (ki % STAGES) * STAGE;
        const char* Bs_ = As_ + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            rg_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm0 + i * 32 + l31;
                const int f = (r >> 1) & 7, c = 4 * ks + 2 * lhi;
                const rg_f4 y0 = *reinterpret_cast<const rg_f4*>(As_ + r * 128 + ((c ^ f) << 4));
                const rg_f4 y1 = *reinterpret_cast<const rg_f4*>(As_ + r * 128 + (((c + 1) ^ f) << 4));
                const float4 x0 = make_float4(y0[0], y0[1], y0[2], y0[3]), x1 = make_float4(y1[0], y1[1], y1[2], y1[3]);
                rg_split8(x0, x1, ah[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = wn0 + j * 32 + l31;
                const int off = r * 64 + (((2 * ks + lhi) ^ ((r >> 2) & 3)) << 4);
                bh[j] = __builtin_bit_cast(rg_bf16x8, *reinterpret_cast<const rg_u4*>(Bs_ + off));
                if (P == 2) bl[j] = __builtin_bit_cast(rg_bf16x8, *reinterpret_cast<const rg_u4*>(Bs_ + BN * 64 + off));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (P == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                }
        }
    }

    // ---- epilogue (gemm_bp.hip's): a lane holds, per register quad q, four consecutive output columns of row m = lane & 31
    const bool split = (EPI == REPI_PLAIN) && p.splits > 1;
    float* slab = split ? p.slab + (long)blockIdx.y * p.M * p.N : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + wm0 + i * 32 + l31;
        if (row >= p.M) continue;
        float rsc = 1.f;
        if (EPI == REPI_FULL) rsc = p.e_rowscale ? p.e_rowscale[row / p.e_rows_per_scale] : 1.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + wn0 + j * 32 + 8 * q + 4 * lhi;
                if (col >= p.N) continue;
                float4 v = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (split) { *reinterpret_cast<float4*>(slab + (long)row * p.N + col) = v; continue; }
                if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                const uint32_t didx = (uint32_t)((long)row * p.N + col);
                if (EPI == REPI_PLAIN) {
                    if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(p.C + (long)row * p.ldc + col); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                }
                if (EPI == REPI_GELU) {
                    if (p.U) *reinterpret_cast<float4*>(p.U + (long)row * p.ldu_out + col) = v;
                    v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
                }
                if (EPI == REPI_DGELU) {
                    const float4 u4 = *reinterpret_cast<const float4*>(p.gelu_u + (long)row * p.ldu + col);
                    v.x *= gelu_grad_f(u4.x); v.y *= gelu_grad_f(u4.y); v.z *= gelu_grad_f(u4.z); v.w *= gelu_grad_f(u4.w);
                }
                if (EPI != REPI_PLAIN && p.e_drop) {
                    const float4 ds = mdvit_drop_scale4(ek0, ek1, didx, p.e_thresh, p.e_inv_keep);
                    v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
                }
                if (EPI == REPI_FULL) {
                    v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
                    if (p.residual) {
                        const float4 r4 = *reinterpret_cast<const float4*>(p.residual + (long)row * p.ldr + col);
                        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                    }
                }
                if (p.C) *reinterpret_cast<float4*>(p.C + (long)row * p.ldc + col) = v;
                if (p.Cp) {
                    uint2 hi, lo;
                    mdvit_split_bf16x3(v, hi, lo);
                    uint16_t* d = p.Cp + (long)row * p.ldcp + col;
                    *reinterpret_cast<uint2*>(d) = hi;
                    if (P == 2) *reinterpret_cast<uint2*>(d + p.c_plane) = lo;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// TN (weight gradient): C[M,N] (+)= sum_k A[k][m] B[k][n]
// ---------------------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int P, int STAGES>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64) void gemm_ring_tn_kernel(RingArgs p) {
    constexpr int WAVES_N = BN / 64, NW = (BM / 64) * WAVES_N, NT = NW * 64;
    constexpr int A_BYTES = BK * BM * 4, B_BYTES = BK * BN * 4, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;
    static_assert(NA % NW == 0 && NB % NW == 0, "pieces must deal evenly to the waves (counted vmcnt)");
    constexpr int G = (NA + NB) / NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tile = rg_xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg) / BK;
    const int wm0 = (wave / WAVES_N) * 64, wn0 = (wave % WAVES_N) * 64;
    const float* Bf = reinterpret_cast<const float*>(p.B);
    // a piece = 1 KiB = 256 floats of the [32][BM] image: rows of BM floats; lane i holds floats 4i..4i+3 of the piece
    constexpr int A_RPP = 256 / BM, B_RPP = 256 / BN;                   // token rows per piece (BM, BN <= 256)
    auto issue = [&](int ki, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE;
        const int k0 = kbeg + ki * BK;
#pragma unroll
        for (int q0 = 0; q0 < NA / NW; ++q0) {
            const int q = q0 * NW + wave;
            const int e = lane * 4, row = q * A_RPP + e / BM;
            int col = m0 + e % BM;
            col = col < p.M ? col : p.M - 4;                             // past the edge: valid memory, columns the epilogue never stores
            __builtin_amdgcn_global_load_lds(p.A + (long)(k0 + row) * p.lda + col, base + q * 1024, 16, 0, 0);
        }
#pragma unroll
        for (int q0 = 0; q0 < NB / NW; ++q0) {
            const int q = q0 * NW + wave;
            const int e = lane * 4, row = q * B_RPP + e / BN;
            int col = n0 + e % BN;
            col = col < p.N ? col : p.N - 4;
            __builtin_amdgcn_global_load_lds(Bf + (long)(k0 + row) * p.ldb + col, base + A_BYTES + q * 1024, 16, 0, 0);
        }
    };

    rg_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_cs = p.colsum != nullptr && tn == 0 && tid < BM;
    float cs = 0.f;

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue(s, s);
    for (int ki = 0; ki < nk; ++ki) {
        const int ahead = min(nk - 1 - ki, STAGES - 2);
        if (STAGES >= 4 && ahead >= 2) rg_wait_vmcnt<2 * G>();
        else if (ahead >= 1) rg_wait_vmcnt<G>();
        else rg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (ki + STAGES - 1 < nk) issue(ki + STAGES - 1, (ki + STAGES - 1) % STAGES);
        const float* As_ = reinterpret_cast<const float*>(smem + (ki % STAGES) * STAGE);
        const float* Bs_ = As_ + BK * BM;
        if (do_cs) {
#pragma unroll
            for (int t = 0; t < BK; ++t) cs += As_[t * BM + tid];
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int t0 = 16 * ks + 8 * lhi;
            rg_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float* src = As_ + t0 * BM + wm0 + i * 32 + l31;
                const float4 x0 = make_float4(src[0], src[BM], src[2 * BM], src[3 * BM]);
                const float4 x1 = make_float4(src[4 * BM], src[5 * BM], src[6 * BM], src[7 * BM]);
                rg_split8(x0, x1, ah[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* src = Bs_ + t0 * BN + wn0 + j * 32 + l31;
                const float4 x0 = make_float4(src[0], src[BN], src[2 * BN], src[3 * BN]);
                const float4 x1 = make_float4(src[4 * BN], src[5 * BN], src[6 * BN], src[7 * BN]);
                rg_split8(x0, x1, bh[j], bl[j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (P == 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                }
        }
    }
    if (do_cs && m0 + tid < p.M) atomicAdd(&p.colsum[m0 + tid], cs);

    const bool split = p.splits > 1;
    float* slab = split ? p.slab + (long)blockIdx.y * p.M * p.N : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + wm0 + i * 32 + l31;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + wn0 + j * 32 + 8 * q + 4 * lhi;
                if (col >= p.N) continue;
                float4 v = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (split) { *reinterpret_cast<float4*>(slab + (long)row * p.N + col) = v; continue; }
                float* dst = p.C + (long)row * p.ldc + col;
                if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *reinterpret_cast<float4*>(dst) = v;
            }
        }
    }
}

template <typename K>
int rg_set_lds(K kernel, int bytes) {
    static int done[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (done[dev] >= bytes) return MDVIT_OK;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "ring gemm: cannot raise the dynamic LDS limit to %d: %s", bytes, hipGetErrorString(e));
    done[dev] = bytes;
    return MDVIT_OK;
}

template <int BM, int BN, int P, int STAGES, int EPI>
int launch_ring_nt(const RingArgs& a, hipStream_t s) {
    constexpr int smem = STAGES * (BM * 128 + P * BN * 64);
    auto k = gemm_ring_nt_kernel<BM, BN, P, STAGES, EPI>;
    const int rc = rg_set_lds(k, smem);
    if (rc != MDVIT_OK) return rc;
    hipLaunchKernelGGL(k, dim3(a.tiles_m * a.tiles_n, a.splits), dim3((BM / 64) * (BN / 64) * 64), smem, s, a);
    return MDVIT_OK;
}

template <int BM, int BN, int P, int STAGES>
int launch_ring_nt_epi(const RingArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case REPI_PLAIN: return launch_ring_nt<BM, BN, P, STAGES, REPI_PLAIN>(a, s);
        case REPI_GELU: return launch_ring_nt<BM, BN, P, STAGES, REPI_GELU>(a, s);
        case REPI_DGELU: return launch_ring_nt<BM, BN, P, STAGES, REPI_DGELU>(a, s);
        case REPI_FULL: return launch_ring_nt<BM, BN, P, STAGES, REPI_FULL>(a, s);
    }
    return mdvit_set_error(MDVIT_E_SHAPE, "ring gemm: epilogue %d is not built", epi);
}

template <int BM, int BN, int P, int STAGES>
int launch_ring_tn(const RingArgs& a, hipStream_t s) {
    constexpr int smem = STAGES * BK * (BM + BN) * 4;
    auto k = gemm_ring_tn_kernel<BM, BN, P, STAGES>;
    const int rc = rg_set_lds(k, smem);
    if (rc != MDVIT_OK) return rc;
    hipLaunchKernelGGL(k, dim3(a.tiles_m * a.tiles_n, a.splits), dim3((BM / 64) * (BN / 64) * 64), smem, s, a);
    return MDVIT_OK;
}

}  // namespace

int mdvit_gemm_splitk_reduce(const float* slab, const float* bias, float* C, long ldc, int M, int N, int splits, int accumulate, hipStream_t s);

int g_ring_nt_cfg = -1;       // tuning hook: -1 planner, 0: 256x128, 1: 128x128
int g_ring_enable = 1;

// Library-internal: NT with fp32 A and plane B (called by mdvit_gemm_planes when its planner prefers the ring).  cfg 0: 256x128 (8 waves,
// 3 x 48 KB), cfg 1: 128x128 (4 waves, 4 x 32 KB).  Returns -1 if the shape is not covered.
int mdvit_ring_nt(const MdvitPlaneGemmDesc* d, int cfg, int splits, int kps, hipStream_t s) {
    RingArgs a;
    memset(&a, 0, sizeof(a));
    a.A = (const float*)d->A; a.lda = d->lda; a.B = d->B; a.ldb = d->ldb; a.b_plane = d->b_plane;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.C = d->C; a.ldc = d->ldc; a.Cp = (uint16_t*)d->Cp; a.ldcp = d->ldcp; a.c_plane = d->c_plane; a.U = d->U; a.ldu_out = d->ldu_out;
    a.bias = d->bias;
    a.e_drop = d->e_drop_p > 0.f; a.e_k0 = d->e_key0; a.e_k1 = d->e_key1;
    a.e_thresh = (uint32_t)((double)d->e_drop_p * 4294967296.0); a.e_inv_keep = 1.f / (1.f - d->e_drop_p);
    a.e_rowscale = d->e_rowscale; a.e_rows_per_scale = d->e_rows_per_scale > 0 ? d->e_rows_per_scale : 1;
    a.residual = d->residual; a.ldr = d->ldr; a.gelu_u = d->gelu_u; a.ldu = d->ldu;
    a.accumulate = d->accumulate; a.seed = d->drop_seed;
    a.splits = splits; a.k_per_split = kps; a.slab = (float*)d->ws;
    int epi = REPI_PLAIN;
    if (d->epi == MDVIT_EPI_GELU_DUAL) epi = REPI_GELU;
    else if (d->epi == MDVIT_EPI_DGELU) epi = REPI_DGELU;
    else if (a.e_drop || d->e_rowscale || d->residual) epi = REPI_FULL;
    const int BMv = cfg == 0 ? 256 : 128;
    a.tiles_m = cdiv(d->M, BMv); a.tiles_n = cdiv(d->N, 128);
    int rc;
    if (cfg == 0) rc = d->planes == 2 ? launch_ring_nt_epi<256, 128, 2, 3>(a, epi, s) : launch_ring_nt_epi<256, 128, 1, 3>(a, epi, s);
    else rc = d->planes == 2 ? launch_ring_nt_epi<128, 128, 2, 4>(a, epi, s) : launch_ring_nt_epi<128, 128, 1, 4>(a, epi, s);
    return rc;
}

// Library-internal: TN weight gradient from fp32 token-major operands (called by mdvit_gemm_f32 for precision >= 1).
int mdvit_ring_tn(const MdvitGemmDesc* d, int planes, int splits, int kps, hipStream_t s) {
    RingArgs a;
    memset(&a, 0, sizeof(a));
    a.A = d->A; a.lda = d->lda; a.B = d->B; a.ldb = d->ldb;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.C = d->C; a.ldc = d->ldc;
    a.accumulate = d->accumulate; a.colsum = d->colsum_a;
    a.splits = splits; a.k_per_split = kps; a.slab = (float*)d->ws;
    a.tiles_m = cdiv(d->M, 128); a.tiles_n = cdiv(d->N, 128);
    return planes == 2 ? launch_ring_tn<128, 128, 2, 4>(a, s) : launch_ring_tn<128, 128, 1, 4>(a, s);
}

extern "C" int mdvit_gemm_ring_config(int32_t enable, int32_t nt_cfg) {
    g_ring_enable = enable ? 1 : 0;
    g_ring_nt_cfg = (nt_cfg >= 0 && nt_cfg <= 1) ? nt_cfg : -1;
    return MDVIT_OK;
}
