// fp32 GEMM family on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit an fmaf chain, 157 TF peak).  One kernel template covers the forward linear layers
// (NT), dgrad (NN) and wgrad (TN, split along the token axis with fp32 atomics), with the
// elementwise neighbours of each GEMM fused in:
//   A-prologue : dropout-mask x per-sample DropPath scale applied to A while it is staged
//                (the backward of  res + droppath(dropout(.)) ).
//   epilogue   : +bias | exact-erf GELU (dual store u, h=dropout(gelu(u))) | dropout |
//                DropPath row scale | +residual | x gelu'(u) x dropout-mask (fc2 dgrad).
// Replaces nn.Linear / 1x1 nn.Conv2d / einsum call sites of the reference:
//   mdvit.py:288 (qkv), :310-311 (proj+drop), mpvit.py:71-78 (Mlp), Decoders.py:196,319-331 (1x1 convs).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int NTHREADS = 256;

struct GemmArgs {
    const float* A; const float* B; float* C; float* C2;
    long lda, ldb, ldc;
    int M, N, K;
    const float* bias;
    // A prologue
    int a_drop; uint32_t a_k0, a_k1, a_thresh; float a_inv_keep;
    const float* a_rowscale; int a_rows_per_scale;
    // epilogue
    int epi;                       // MDVIT_EPI_*
    int e_drop; uint32_t e_k0, e_k1, e_thresh; float e_inv_keep;
    const float* e_rowscale; int e_rows_per_scale;
    const float* residual; long ldr;
    const float* gelu_u; long ldu;
    int splits; int k_per_split;   // split along K (atomics) when splits > 1
    int tiles_m, tiles_n;
};

// Bijective XCD-aware remap (guide T1): consecutive logical tiles share an XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool TA, bool TB>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(GemmArgs p) {
    // k-contiguous operands are transposed on the LDS write: odd leading dimension -> conflict-free ds_write_b32;
    // m/n-contiguous operands are written as float4: leading dimension % 4 == 0.
    constexpr int LDSA = TA ? BM + 4 : BM + 1, LDSB = TB ? BN + 1 : BN + 4;
    constexpr int WTM = BM / WAVES_M / 32, WTN = BN / WAVES_N / 32;   // 32x32 blocks per wave
    constexpr int A_V4 = BM * BK / 4 / NTHREADS, B_V4 = BN * BK / 4 / NTHREADS;
    constexpr int KT = BK / 4;                                        // threads per k-contiguous row
    __shared__ __attribute__((aligned(16))) float smem[BK * LDSA + BK * LDSB];
    float* As = smem;
    float* Bs = smem + BK * LDSA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int wm0 = (wave / WAVES_N) * (BM / WAVES_M), wn0 = (wave % WAVES_N) * (BN / WAVES_N);

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_V4], rb[B_V4];

    auto load_a = [&](int k0) {
#pragma unroll
        for (int v = 0; v < A_V4; ++v) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!TA) {   // A[m][k], k contiguous: BK/4 threads per row
                const int r = tid / KT + v * (NTHREADS / KT), m = m0 + r, k = k0 + (tid % KT) * 4;
                if (m < p.M && k < kend) {
                    x = *reinterpret_cast<const float4*>(p.A + (long)m * p.lda + k);
                    if (p.a_drop | (p.a_rowscale != nullptr)) {
                        float rs = p.a_rowscale ? p.a_rowscale[m / p.a_rows_per_scale] : 1.f;
                        if (p.a_drop) {
                            const uint32_t idx = (uint32_t)((long)m * p.K + k);
                            x.x *= rs * mdvit_drop_scale(p.a_k0, p.a_k1, idx, p.a_thresh, p.a_inv_keep);
                            x.y *= rs * mdvit_drop_scale(p.a_k0, p.a_k1, idx + 1, p.a_thresh, p.a_inv_keep);
                            x.z *= rs * mdvit_drop_scale(p.a_k0, p.a_k1, idx + 2, p.a_thresh, p.a_inv_keep);
                            x.w *= rs * mdvit_drop_scale(p.a_k0, p.a_k1, idx + 3, p.a_thresh, p.a_inv_keep);
                        } else { x.x *= rs; x.y *= rs; x.z *= rs; x.w *= rs; }
                    }
                }
            } else {     // A stored [k][m], m contiguous (wgrad: A = dY^T)
                constexpr int TPR = BM / 4;                 // threads per k-row
                const int kk = tid / TPR + v * (NTHREADS / TPR), k = k0 + kk, m = m0 + (tid % TPR) * 4;
                if (k < kend && m < p.M) {
                    x = *reinterpret_cast<const float4*>(p.A + (long)k * p.lda + m);
                    if (p.a_drop | (p.a_rowscale != nullptr)) {
                        // the stored tensor is [k][m] = dY[token k][feature m]; mask index = k*M + m
                        float rs = p.a_rowscale ? p.a_rowscale[k / p.a_rows_per_scale] : 1.f;
                        if (p.a_drop) {
                            const uint32_t idx = (uint32_t)((long)k * p.M + m);
                            x.x *= rs * mdvit_drop_scale(p.a_k0, p.a_k1, idx, p.a_thresh, p.a_inv_keep);
                            x.y *= rs * mdvit_drop_scale(p.a_k0, p.a_k1, idx + 1, p.a_thresh, p.a_inv_keep);
                            x.z *= rs * mdvit_drop_scale(p.a_k0, p.a_k1, idx + 2, p.a_thresh, p.a_inv_keep);
                            x.w *= rs * mdvit_drop_scale(p.a_k0, p.a_k1, idx + 3, p.a_thresh, p.a_inv_keep);
                        } else { x.x *= rs; x.y *= rs; x.z *= rs; x.w *= rs; }
                    }
                }
            }
            ra[v] = x;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int v = 0; v < B_V4; ++v) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TB) {    // B[n][k], k contiguous (weights as stored by nn.Linear)
                const int r = tid / KT + v * (NTHREADS / KT), n = n0 + r, k = k0 + (tid % KT) * 4;
                if (n < p.N && k < kend) x = *reinterpret_cast<const float4*>(p.B + (long)n * p.ldb + k);
            } else {     // B[k][n], n contiguous
                constexpr int TPR = BN / 4;
                const int kk = tid / TPR + v * (NTHREADS / TPR), k = k0 + kk, n = n0 + (tid % TPR) * 4;
                if (k < kend && n < p.N) x = *reinterpret_cast<const float4*>(p.B + (long)k * p.ldb + n);
            }
            rb[v] = x;
        }
    };
    auto store_smem = [&]() {
#pragma unroll
        for (int v = 0; v < A_V4; ++v) {
            if (!TA) {
                const int r = tid / KT + v * (NTHREADS / KT), c = (tid % KT) * 4;
                As[(c + 0) * LDSA + r] = ra[v].x; As[(c + 1) * LDSA + r] = ra[v].y;
                As[(c + 2) * LDSA + r] = ra[v].z; As[(c + 3) * LDSA + r] = ra[v].w;
            } else {
                constexpr int TPR = BM / 4;
                const int kk = tid / TPR + v * (NTHREADS / TPR), c = (tid % TPR) * 4;
                *reinterpret_cast<float4*>(&As[kk * LDSA + c]) = ra[v];
            }
        }
#pragma unroll
        for (int v = 0; v < B_V4; ++v) {
            if (TB) {
                const int r = tid / KT + v * (NTHREADS / KT), c = (tid % KT) * 4;
                Bs[(c + 0) * LDSB + r] = rb[v].x; Bs[(c + 1) * LDSB + r] = rb[v].y;
                Bs[(c + 2) * LDSB + r] = rb[v].z; Bs[(c + 3) * LDSB + r] = rb[v].w;
            } else {
                constexpr int TPR = BN / 4;
                const int kk = tid / TPR + v * (NTHREADS / TPR), c = (tid % TPR) * 4;
                *reinterpret_cast<float4*>(&Bs[kk * LDSB + c]) = rb[v];
            }
        }
    };

    load_a(kbeg);
    load_b(kbeg);
    store_smem();
    __syncthreads();
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) { load_a(k0 + BK); load_b(k0 + BK); }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = 2 * kk + lhi;
            float a[WTM], b[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) a[i] = As[krow * LDSA + wm0 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < WTN; ++j) b[j] = Bs[krow * LDSB + wn0 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_smem();
            __syncthreads();
        }
    }

    // ---- epilogue.  The MFMA ran as D = B^T-tile x A-tile, so D[row = n][col = m]: lane holds, for each
    // register quad q = r>>2, FOUR CONSECUTIVE output columns n = 8q + 4*(lane>>5) + (r&3) of output row
    // m = lane&31  ->  one 16-byte store per quad instead of four scalar stores.
    const bool atomic = p.splits > 1;
    const bool add_bias = p.bias != nullptr && blockIdx.y == 0;
    const bool vec = ((p.N & 3) == 0) && ((p.ldc & 3) == 0);
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
        const int row = m0 + wm0 + i * 32 + l31;
        if (row >= p.M) continue;
        const float rsc = p.e_rowscale ? p.e_rowscale[row / p.e_rows_per_scale] : 1.f;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + wn0 + j * 32 + 8 * q + 4 * lhi;
                if (col >= p.N) continue;
                float v[4] = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                const int nv = min(4, p.N - col);
                if (add_bias) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (t < nv) v[t] += p.bias[col + t];
                }
                float* dst = p.C + (long)row * p.ldc + col;
                if (atomic) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (t < nv) atomicAdd(dst + t, v[t]);
                    continue;
                }
                const uint32_t didx = (uint32_t)((long)row * p.N + col);
                if (p.epi == MDVIT_EPI_GELU_DUAL) {
                    float h[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        h[t] = gelu_f(v[t]);
                        if (p.e_drop) h[t] *= mdvit_drop_scale(p.e_k0, p.e_k1, didx + t, p.e_thresh, p.e_inv_keep);
                    }
                    float* dst2 = p.C2 + (long)row * p.ldc + col;
                    if (vec) {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(dst2) = make_float4(h[0], h[1], h[2], h[3]);
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (t < nv) { dst[t] = v[t]; dst2[t] = h[t]; }
                    }
                    continue;
                }
                if (p.epi == MDVIT_EPI_DGELU) {
                    const float* up = p.gelu_u + (long)row * p.ldu + col;
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (t < nv) v[t] *= gelu_grad_f(up[t]);
                }
                if (p.e_drop) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] *= mdvit_drop_scale(p.e_k0, p.e_k1, didx + t, p.e_thresh, p.e_inv_keep);
                }
                if (p.e_rowscale) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] *= rsc;
                }
                if (p.residual) {
                    const float* rp = p.residual + (long)row * p.ldr + col;
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (t < nv) v[t] += rp[t];
                }
                if (vec) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (t < nv) dst[t] = v[t];
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmArgs& a, int ta, int tb, hipStream_t s) {
    dim3 grid(a.tiles_m * a.tiles_n, a.splits), block(NTHREADS);
    if (!ta && tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, a);
    else if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, s, a);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, s, a);
    return 0;
}

}  // namespace

extern "C" int mdvit_gemm_f32(const MdvitGemmDesc* d, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(d != nullptr, MDVIT_E_SHAPE, "gemm: null descriptor");
    MDVIT_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, MDVIT_E_SHAPE, "gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    MDVIT_CHECK_ARG(d->A && d->B && d->C, MDVIT_E_SHAPE, "gemm: null operand");
    MDVIT_CHECK_ARG(aligned16(d->A) && aligned16(d->B) && (d->lda % 4 == 0) && (d->ldb % 4 == 0), MDVIT_E_ALIGN,
                    "gemm: operands must be 16-byte aligned with leading dimensions %% 4 == 0 (lda=%ld ldb=%ld)", d->lda, d->ldb);
    MDVIT_CHECK_ARG(d->trans_a ? (d->M % 4 == 0) : (d->K % 4 == 0), MDVIT_E_ALIGN, "gemm: contiguous extent of A must be %% 4 (M=%d K=%d ta=%d)", d->M, d->K, d->trans_a);
    MDVIT_CHECK_ARG(d->trans_b ? (d->K % 4 == 0) : (d->N % 4 == 0), MDVIT_E_ALIGN, "gemm: contiguous extent of B must be %% 4 (N=%d K=%d tb=%d)", d->N, d->K, d->trans_b);
    MDVIT_CHECK_ARG(d->epi != MDVIT_EPI_GELU_DUAL || d->C2, MDVIT_E_SHAPE, "gemm: GELU_DUAL needs C2");
    MDVIT_CHECK_ARG(d->epi != MDVIT_EPI_DGELU || d->gelu_u, MDVIT_E_SHAPE, "gemm: DGELU needs gelu_u");
    MDVIT_CHECK_ARG((long)d->M * d->N < (1L << 32) && (long)d->M * d->K < (1L << 32), MDVIT_E_SHAPE, "gemm: dropout index space exceeds 2^32");

    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = d->A; a.B = d->B; a.C = d->C; a.C2 = d->C2;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.M = d->M; a.N = d->N; a.K = d->K;
    a.bias = d->bias;
    a.a_drop = d->a_drop_p > 0.f; a.a_k0 = d->a_key0; a.a_k1 = d->a_key1;
    a.a_thresh = (uint32_t)((double)d->a_drop_p * 4294967296.0); a.a_inv_keep = 1.f / (1.f - d->a_drop_p);
    a.a_rowscale = d->a_rowscale; a.a_rows_per_scale = d->a_rows_per_scale > 0 ? d->a_rows_per_scale : 1;
    a.epi = d->epi;
    a.e_drop = d->e_drop_p > 0.f; a.e_k0 = d->e_key0; a.e_k1 = d->e_key1;
    a.e_thresh = (uint32_t)((double)d->e_drop_p * 4294967296.0); a.e_inv_keep = 1.f / (1.f - d->e_drop_p);
    a.e_rowscale = d->e_rowscale; a.e_rows_per_scale = d->e_rows_per_scale > 0 ? d->e_rows_per_scale : 1;
    a.residual = d->residual; a.ldr = d->ldr; a.gelu_u = d->gelu_u; a.ldu = d->ldu;

    // tile shape: 256x64 when the output is narrow or N is an odd multiple of 64, else 128x128
    const bool narrow = (d->N <= 64) || (d->N % 128 != 0 && d->N % 64 == 0);
    const int BM = narrow ? 256 : 128, BN = narrow ? 64 : 128;
    a.tiles_m = cdiv(d->M, BM); a.tiles_n = cdiv(d->N, BN);
    const long tiles = (long)a.tiles_m * a.tiles_n;
    int splits = 1;
    const bool plain = d->epi == MDVIT_EPI_NONE && !a.e_drop && !d->e_rowscale && !d->residual;
    if (d->allow_split && plain && tiles < 512 && d->K >= 2048) {
        splits = (int)((1024 + tiles - 1) / tiles);
        const int max_splits = d->K / 512;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
    }
    int kps = cdiv(cdiv(d->K, splits), BK) * BK;
    splits = cdiv(d->K, kps);
    a.splits = splits; a.k_per_split = kps;
    if (splits > 1) {
        MDVIT_CHECK_ARG(d->ldc == d->N, MDVIT_E_SHAPE, "gemm: split reduction needs a dense output (ldc == N)");
        MDVIT_ZERO(d->C, sizeof(float) * (size_t)d->M * d->N, s);
    }
    if (narrow) launch_cfg<256, 64, 4, 1>(a, d->trans_a, d->trans_b, s);
    else launch_cfg<128, 128, 2, 2>(a, d->trans_a, d->trans_b, s);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
