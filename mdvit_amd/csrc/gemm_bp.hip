// "Plane" GEMM family: operands arrive ALREADY split into bf16 planes (hi = RNE bf16(x), lo = RNE bf16(x - hi); a tensor is
// [planes][rows][ld] bf16, hi plane first), so the main loop carries no conversion VALU at all: operand tiles go from HBM / L2
// straight into LDS with global_load_lds (16 bytes per lane, 1 KiB per wave-instruction) and from there into MFMA fragments with
// ds_read_b128.  Weights are split once per optimizer step (mdvit_split_planes_many), activations by the kernel that produces
// them (LayerNorm, attention output, the masked upstream gradient, this GEMM's own epilogues).
//   planes = 2 ("bf16x3"): hi*lo + lo*hi + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate -- same products, same order,
//                          bit-identical to the split-while-staging kernel of gemm.hip;
//   planes = 1 ("bf16")  : hi*hi only -- the bf16 speed mode (half the operand bytes, a third of the MFMAs).
// NT:  C[M,N] = A[M,K] B[N,K]^T   forward linear layers; data gradients against the cached W^T planes
//      (A may also be fp32 -- a_f32 -- and is then split while staged, for producers that do not write planes yet)
// Epilogues as in gemm.hip (bias | GELU | x gelu'(u) x mask with u read or recomputed | dropout + DropPath + residual), each able to
// write its result as fp32 and / or as bf16 planes (the next GEMM's operand).
// LDS image of one operand slab: [plane][row][4 x 16 B] with the 16-byte chunk index XOR-swizzled by (row >> 2) & 3 --
// global_load_lds writes lane-linear (base + lane * 16), so the swizzle is applied to each lane's SOURCE address and again by
// the fragment reads; with it the four 16-lane groups of a ds_read_b128 hit 16 distinct bank slots (64-byte rows, no padding).
// Replaces the same reference call sites as gemm.hip: mdvit.py:288,310-311, mpvit.py:71-78, Decoders.py:196,319-331.
#include "common.h"
#include "gemm_bp.h"

typedef float bp_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bp_bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BK = 32;            // bf16 elements per K slab (64-byte rows)
constexpr int NT_THREADS = 256;


__device__ __forceinline__ int bp_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// One operand slab (ROWS rows x 64 B per plane) from HBM / L2 into LDS as P * ROWS / 16 pieces of 1 KiB, dealt round-robin to the
// 4 waves: piece q covers rows 16q .. 16q+15; lane i lands at row 16q + (i >> 2), physical chunk i & 3, and fetches logical
// chunk (i & 3) ^ ((row >> 2) & 3) of that row (prow / pchunk, precomputed per lane).
template <int ROWS, int P>
__device__ __forceinline__ void bp_glds_slab(const uint16_t* __restrict__ src, long ld, long plane_stride, int row0, int row_limit, int k0, char* dst,
                                             int wave, int prow, int pchunk) {
    constexpr int PER_PLANE = ROWS / 16, NP = P * PER_PLANE;
#pragma unroll
    for (int q0 = 0; q0 < (NP + 3) / 4; ++q0) {
        const int q = q0 * 4 + wave;
        if (NP % 4 == 0 || q < NP) {
            const int pl = q / PER_PLANE, rq = q % PER_PLANE;
            int row = row0 + rq * 16 + prow;
            row = row < row_limit ? row : row_limit - 1;                     // rows past the edge: any valid row (the epilogue discards them)
            const uint16_t* g = src + pl * plane_stride + (long)row * ld + k0 + pchunk * 8;
            __builtin_amdgcn_global_load_lds(g, dst + pl * ROWS * 64 + rq * 1024, 16, 0, 0);
        }
    }
}

// C = A B^T.  BM x BN tile, 4 waves as 2 x 2, each wave (BM/2) x (BN/2) = WTM x WTN blocks of 32 x 32.
// P = planes (2: bf16x3, 1: bf16).  AF32: A is fp32 in HBM, split while staged (register path); otherwise every slab is a
// global_load_lds copy.  Double-buffered LDS, ONE barrier per K slab: slab k+1 is in flight while slab k is multiplied.
template <int BM, int BN, int P, bool AF32, int EPI>
__global__ __launch_bounds__(NT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 8))) void gemm_bp_nt_kernel(BpArgs p) {
    constexpr int WTM = BM / 64, WTN = BN / 64;
    constexpr int A_BYTES = P * BM * 64, B_BYTES = P * BN * 64;          // one slab of each operand
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr bool RC = EPI == BEPI_DGELU_RC;
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t ek0 = p.e_k0 ^ s0, ek1 = p.e_k1 + s1;
    const int tile = bp_xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = RC ? 0 : blockIdx.y * p.k_per_split;
    const int kend = RC ? p.rc_k + p.K : min(p.K, kbeg + p.k_per_split);
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    // ---- staging: a slab of R rows x 64 B per plane is R/16 pieces of 1 KiB; piece q covers rows 16q .. 16q+15, lane i
    // lands at row 16q + (i >> 2), physical chunk i & 3, and fetches logical chunk (i & 3) ^ ((row >> 2) & 3) of that row
    const int prow = lane >> 2, pchunk = (lane & 3) ^ ((lane >> 4) & 3);           // (row>>2)&3 = (prow>>2)&3 since 16q % 16 == 0
    // AF32: 4 threads per row and 16-byte chunk pair: thread loads two float4 (8 k) = one 16-byte bf16 chunk per plane
    constexpr int A_V8 = BM * 4 / NT_THREADS;           // 8-element chunks per thread per slab
    float4 ra[AF32 ? A_V8 : 1][2];
    auto load_a_f32 = [&](int k0) __attribute__((always_inline)) {
        if constexpr (AF32) {
            const float* Af = reinterpret_cast<const float*>(p.A);
#pragma unroll
            for (int v = 0; v < A_V8; ++v) {
                const int idx = tid + v * NT_THREADS, row = idx >> 2, c = idx & 3;
                int m = m0 + row; m = m < p.M ? m : p.M - 1;
                const float* g = Af + (long)m * p.lda + k0 + c * 8;
                ra[v][0] = *reinterpret_cast<const float4*>(g);
                ra[v][1] = *reinterpret_cast<const float4*>(g + 4);
            }
        }
    };
    auto store_a_f32 = [&](char* dst) __attribute__((always_inline)) {
        if constexpr (AF32) {
#pragma unroll
            for (int v = 0; v < A_V8; ++v) {
                const int idx = tid + v * NT_THREADS, row = idx >> 2, c = idx & 3;
                uint2 h0, l0, h1, l1;
                mdvit_split_bf16x3(ra[v][0], h0, l0);
                mdvit_split_bf16x3(ra[v][1], h1, l1);
                const int off = row * 64 + ((c ^ ((row >> 2) & 3)) * 16);
                *reinterpret_cast<uint4*>(dst + off) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                if (P == 2) *reinterpret_cast<uint4*>(dst + BM * 64 + off) = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }
    };
    auto issue = [&](int k0, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE;
        if constexpr (RC) {
            if (k0 < p.rc_k) {
                bp_glds_slab<BM, P>(p.rc_a, p.rc_lda, p.rc_a_plane, m0, p.M, k0, base, wave, prow, pchunk);
                bp_glds_slab<BN, P>(p.rc_b, p.rc_ldb, p.rc_b_plane, n0, p.N, k0, base + A_BYTES, wave, prow, pchunk);
                return;
            }
            k0 -= p.rc_k;
        }
        if constexpr (AF32) load_a_f32(k0);
        else bp_glds_slab<BM, P>(reinterpret_cast<const uint16_t*>(p.A), p.lda, p.a_plane, m0, p.M, k0, base, wave, prow, pchunk);
        bp_glds_slab<BN, P>(p.B, p.ldb, p.b_plane, n0, p.N, k0, base + A_BYTES, wave, prow, pchunk);
    };

    bp_f32x16 acc[WTM][WTN], uacc[RC ? WTM : 1][RC ? WTN : 1];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if (RC) uacc[RC ? i : 0][RC ? j : 0][r] = 0.f; }

    auto mma_into = [&](const char* base, auto& Cacc) __attribute__((always_inline)) {
        const char* As_ = base; const char* Bs_ = base + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = 2 * ks + lhi;                                         // logical 16-byte chunk of this lane's 8 k values
            bp_bf16x8 ah[WTM], al[WTM], bh[WTN], bl[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                const int r = wm0 + i * 32 + l31;
                const int off = r * 64 + ((c ^ ((r >> 2) & 3)) * 16);
                ah[i] = __builtin_bit_cast(bp_bf16x8, *reinterpret_cast<const uint4*>(As_ + off));
                if (P == 2) al[i] = __builtin_bit_cast(bp_bf16x8, *reinterpret_cast<const uint4*>(As_ + BM * 64 + off));
            }
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int r = wn0 + j * 32 + l31;
                const int off = r * 64 + ((c ^ ((r >> 2) & 3)) * 16);
                bh[j] = __builtin_bit_cast(bp_bf16x8, *reinterpret_cast<const uint4*>(Bs_ + off));
                if (P == 2) bl[j] = __builtin_bit_cast(bp_bf16x8, *reinterpret_cast<const uint4*>(Bs_ + BN * 64 + off));
            }
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) {
                    if (P == 2) {
                        Cacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], Cacc[i][j], 0, 0, 0);
                        Cacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], Cacc[i][j], 0, 0, 0);
                    }
                    Cacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], Cacc[i][j], 0, 0, 0);
                }
        }
    };

    // Epilogue operands that do not depend on the product, fetched ahead of the K loop for single-block waves
    constexpr bool EPRE = (WTM * WTN == 1) && (EPI == BEPI_DGELU || EPI == BEPI_FULL);
    float4 epre[EPRE ? 4 : 1];
    if (EPRE) {
        const float* src = EPI == BEPI_DGELU ? p.gelu_u : p.residual;
        const long lds_ = EPI == BEPI_DGELU ? p.ldu : p.ldr;
        const int row = m0 + wm0 + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = n0 + wn0 + 8 * q + 4 * lhi;
            epre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src && row < p.M && col < p.N) epre[q] = *reinterpret_cast<const float4*>(src + (long)row * lds_ + col);
        }
    }

    // ---- main loop
    issue(kbeg, 0);
    if constexpr (AF32) { if (!RC || kbeg >= p.rc_k) store_a_f32(smem); }
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's global_load_lds pieces of slab k0 have landed ...
        __syncthreads();                                   // ... and everybody else's: slab k0 has landed in `buf` (vmcnt(0) + barrier); buf^1 is free
        const bool more = k0 + BK < kend;
        if (more) issue(k0 + BK, buf ^ 1);
        if constexpr (RC) {
            if (k0 < p.rc_k) mma_into(smem + buf * STAGE, uacc);
            else mma_into(smem + buf * STAGE, acc);
        } else {
            mma_into(smem + buf * STAGE, acc);
        }
        if constexpr (AF32) {
            if (more && (!RC || k0 + BK >= p.rc_k)) store_a_f32(smem + (buf ^ 1) * STAGE);
        }
        buf ^= 1;
    }

    // ---- epilogue: D[row = n][col = m] per 32x32 block: a lane holds, for each register quad q, FOUR CONSECUTIVE output
    // columns n = 8q + 4*(lane>>5) + (r&3) of output row m = lane&31
    const bool split = (EPI == BEPI_PLAIN) && p.splits > 1;
    float* slab = split ? p.slab + (long)blockIdx.y * p.M * p.N : nullptr;
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
        const int row = m0 + wm0 + i * 32 + l31;
        if (row >= p.M) continue;
        float rsc = 1.f;
        if (EPI == BEPI_FULL) rsc = p.e_rowscale ? p.e_rowscale[row / p.e_rows_per_scale] : 1.f;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + wn0 + j * 32 + 8 * q + 4 * lhi;
                if (col >= p.N) continue;
                float4 v = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (split) { *reinterpret_cast<float4*>(slab + (long)row * p.N + col) = v; continue; }
                if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                const uint32_t didx = (uint32_t)((long)row * p.N + col);
                if (EPI == BEPI_PLAIN) {
                    if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(p.C + (long)row * p.ldc + col); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                }
                if (EPI == BEPI_GELU) {
                    if (p.U) *reinterpret_cast<float4*>(p.U + (long)row * p.ldu_out + col) = v;
                    v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
                }
                if (EPI == BEPI_DGELU) {
                    const float4 u4 = EPRE ? epre[q] : *reinterpret_cast<const float4*>(p.gelu_u + (long)row * p.ldu + col);
                    v.x *= gelu_grad_f(u4.x); v.y *= gelu_grad_f(u4.y); v.z *= gelu_grad_f(u4.z); v.w *= gelu_grad_f(u4.w);
                }
                if (RC) {
                    float4 u4 = make_float4(uacc[RC ? i : 0][RC ? j : 0][4 * q + 0], uacc[RC ? i : 0][RC ? j : 0][4 * q + 1],
                                            uacc[RC ? i : 0][RC ? j : 0][4 * q + 2], uacc[RC ? i : 0][RC ? j : 0][4 * q + 3]);
                    if (p.rc_bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.rc_bias + col); u4.x += b4.x; u4.y += b4.y; u4.z += b4.z; u4.w += b4.w; }
                    v.x *= gelu_grad_f(u4.x); v.y *= gelu_grad_f(u4.y); v.z *= gelu_grad_f(u4.z); v.w *= gelu_grad_f(u4.w);
                }
                if (EPI != BEPI_PLAIN && p.e_drop) {
                    const float4 ds = mdvit_drop_scale4(ek0, ek1, didx, p.e_thresh, p.e_inv_keep);
                    v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
                }
                if (EPI == BEPI_FULL) {
                    v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
                    if (p.residual) {
                        const float4 r4 = EPRE ? epre[q] : *reinterpret_cast<const float4*>(p.residual + (long)row * p.ldr + col);
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

// ---- fp32 [rows, cols] (leading dimension ld) -> bf16 planes, optionally transposed --------------------------------------------
__device__ __forceinline__ void split_tiles(const float* __restrict__ in, long ld_in, uint16_t* __restrict__ out, long ld_out, long plane_stride,
                                            int rows, int cols, int tr, int planes, float (*tile)[33]) {
    const int tiles_x = (cols + 31) / 32, tiles_y = (rows + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tidx = blockIdx.x; tidx < tiles_x * tiles_y; tidx += gridDim.x) {
        const int r0 = (tidx / tiles_x) * 32, c0 = (tidx % tiles_x) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty + 8 * i, c = c0 + tx;
            tile[ty + 8 * i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int orow, ocol; float v;
            if (tr) { ocol = r0 + tx; orow = c0 + ty + 8 * i; v = tile[tx][ty + 8 * i]; if (orow >= cols || ocol >= rows) continue; }
            else { orow = r0 + ty + 8 * i; ocol = c0 + tx; v = tile[ty + 8 * i][tx]; if (orow >= rows || ocol >= cols) continue; }
            uint16_t hi, lo;
            mdvit_split1_bf16x3(v, hi, lo);
            out[(long)orow * ld_out + ocol] = hi;
            if (planes == 2) out[plane_stride + (long)orow * ld_out + ocol] = lo;
        }
        __syncthreads();
    }
}
// many tensors per launch: items [n][8] int64 = {src, dst, ld_src, rows, cols, transpose, ld_dst, plane_stride}; dst rows are `rows`
// long when transposed.  Used once per optimizer step on the weights (W planes for the forward GEMMs, W^T planes for the data gradients).
__global__ __launch_bounds__(256) void split_planes_many_kernel(const long long* __restrict__ items, int planes) {
    __shared__ float tile[32][33];
    const long long* it = items + 8 * (long)blockIdx.y;
    split_tiles(reinterpret_cast<const float*>(it[0]), (long)it[2], reinterpret_cast<uint16_t*>(it[1]), (long)it[6], (long)it[7],
                (int)it[3], (int)it[4], (int)it[5], planes, tile);
}
__global__ __launch_bounds__(256) void split_planes_t_kernel(const float* __restrict__ in, long ld_in, uint16_t* __restrict__ out, long ld_out, long plane_stride,
                                                             int rows, int cols, int tr, int planes) {
    __shared__ float tile[32][33];
    split_tiles(in, ld_in, out, ld_out, plane_stride, rows, cols, tr, planes, tile);
}

// one tensor, no transpose, vectorised: 8 elements per thread
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ in, long ld_in, uint16_t* __restrict__ out, long ld_out, long plane_stride,
                                                           long rows, int cols8, int planes) {
    const long total = rows * cols8;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long r = e / cols8; const int c = (int)(e % cols8) * 8;
        const float4 a = *reinterpret_cast<const float4*>(in + r * ld_in + c), b = *reinterpret_cast<const float4*>(in + r * ld_in + c + 4);
        uint2 h0, l0, h1, l1;
        mdvit_split_bf16x3(a, h0, l0); mdvit_split_bf16x3(b, h1, l1);
        *reinterpret_cast<uint4*>(out + r * ld_out + c) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        if (planes == 2) *reinterpret_cast<uint4*>(out + plane_stride + r * ld_out + c) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

struct BpPlan { int cfg, tiles_m, tiles_n, splits, kps; };
int g_bp_force_cfg = -1, g_bp_force_splits = 0;

// cfg 0: 128x128 (2 workgroups / CU: 64 KB of LDS each)   1: 128x64   2: 64x64 (5 workgroups / CU)
// cfg 3: 256x256, eight phase-split waves, one workgroup per CU (gemm_ph.hip)
BpPlan plan_ph(const MdvitPlaneGemmDesc* d, int cfg) {
    static const int SPLITS[] = {1, 2, 3, 4, 6, 8};
    const int kt = d->planes == 2 ? 32 : 64;
    const bool plain = d->epi == MDVIT_EPI_NONE && !(d->e_drop_p > 0.f) && !d->e_rowscale && !d->residual && !d->Cp;
    const bool can_split = d->allow_split && plain && d->K >= 512;
    const long tm = cdiv(d->M, 256), tn = cdiv(d->N, 256);
    BpPlan best{cfg, (int)tm, (int)tn, 1, d->K};
    double best_cost = 1e300;
    for (int si = 0; si < (int)(sizeof(SPLITS) / sizeof(int)); ++si) {
        const int want = SPLITS[si];
        if (want > 1 && (!can_split || want > d->K / 256)) break;
        if (g_bp_force_splits > 0 && can_split && want != g_bp_force_splits) continue;
        const int kps = cdiv(cdiv(d->K, want), kt) * kt;
        const int splits = cdiv(d->K, kps);
        const double rounds = (double)cdiv(tm * tn * splits, 256);
        double cost = rounds * (256.0 * 256.0 * kps * (d->planes == 2 ? 3.0 : 1.0) / (1024.0 * 4.0) / 0.7 + 6000.0);
        if (splits > 1) cost += 12000.0 + (double)(splits + 1) * d->M * d->N * 8.0 / 1250.0;
        if (cost < best_cost) { best_cost = cost; best = BpPlan{cfg, (int)tm, (int)tn, splits, kps}; }
    }
    return best;
}


BpPlan plan_bp(const MdvitPlaneGemmDesc* d) {
    if (g_bp_force_cfg >= 6) {
        BpPlan pl{g_bp_force_cfg, cdiv(d->M, 128), cdiv(d->N, g_bp_force_cfg == 6 ? 160 : 128), 1, d->K};
        const bool plain_ = d->epi == MDVIT_EPI_NONE && !(d->e_drop_p > 0.f) && !d->e_rowscale && !d->residual && !d->Cp;
        if (g_bp_force_splits > 1 && d->allow_split && plain_) { pl.kps = cdiv(cdiv(d->K, g_bp_force_splits), 32) * 32; pl.splits = cdiv(d->K, pl.kps); }
        return pl;
    }
    if (g_bp_force_cfg >= 3) return plan_ph(d, g_bp_force_cfg);
    const int epi_reads = d->gelu_u != nullptr || d->residual != nullptr || d->accumulate != 0;
    if (g_bp_force_cfg < 0 && !d->rc_a && mdvit_gemm_ph_prefers_epi(d->M, d->N, d->K, d->planes, epi_reads)) {
        BpPlan pl = plan_ph(d, 3);
        if (pl.splits == 1) return pl;
    }
    if (g_bp_force_cfg < 0 && !d->rc_a && !d->Cp) {            // the 128-row phase-split tile (gemm_pm.hip): mid-size products
        const int cfg = mdvit_gemm_pm_prefers(d->M, d->N, d->K, d->planes, d->a_f32);
        if (cfg) return BpPlan{cfg, cdiv(d->M, 128), cdiv(d->N, cfg == 6 ? 160 : 128), 1, d->K};
        // ... and as 2-4 K ranges where its tiles alone would leave most of the chip idle (stage 3 at bs=4); plain products only
        const bool plain_ = d->epi == MDVIT_EPI_NONE && !(d->e_drop_p > 0.f) && !d->e_rowscale && !d->residual;
        const int sp = (d->allow_split && plain_) ? mdvit_gemm_pm_splits(d->M, d->N, d->K, d->planes, d->a_f32) : 1;
        if (sp > 1) {
            const int c2 = d->N % 160 == 0 ? 6 : 7;
            const int kps = cdiv(cdiv(d->K, sp), 32) * 32;
            return BpPlan{c2, cdiv(d->M, 128), cdiv(d->N, c2 == 6 ? 160 : 128), cdiv(d->K, kps), kps};
        }
    }
    static const int BMs[3] = {128, 128, 64}, BNs[3] = {128, 64, 64}, OCC[3] = {2, 3, 5};
    static const int SPLITS[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    const bool plain = d->epi == MDVIT_EPI_NONE && !(d->e_drop_p > 0.f) && !d->e_rowscale && !d->residual && !d->Cp;
    const bool can_split = d->allow_split && plain && d->K >= 512;
    BpPlan best{0, cdiv(d->M, 128), cdiv(d->N, 128), 1, cdiv(d->K, BK) * BK};
    if (d->rc_a) best = BpPlan{1, cdiv(d->M, 128), cdiv(d->N, 64), 1, cdiv(d->K, BK) * BK};
    double best_cost = 1e300;
    for (int c = 0; c < 3; ++c) {
        if (d->rc_a && c == 0) continue;              // two accumulator sets: the 128x128 tile would spill
        if (g_bp_force_cfg >= 0 && c != g_bp_force_cfg) continue;
        const long tm = cdiv(d->M, BMs[c]), tn = cdiv(d->N, BNs[c]);
        for (int si = 0; si < (int)(sizeof(SPLITS) / sizeof(int)); ++si) {
            const int want = SPLITS[si];
            if (want > 1 && (!can_split || want > d->K / 256)) break;
            if (g_bp_force_splits > 0 && can_split && want != g_bp_force_splits) continue;
            const int kps = cdiv(cdiv(d->K, want), BK) * BK;
            const int splits = cdiv(d->K, kps);
            const long wgs = tm * tn * splits;
            const double slots = 256.0 * OCC[c];
            const double rounds = wgs <= slots ? 1.0 : (double)wgs / slots;
            // per workgroup: MFMA cycles (planes==2: 3 per product) at 4 SIMDs + a fixed per-slab latency + prologue / epilogue
            const double mf = (d->planes == 2 ? 3.0 : 1.0) * BMs[c] * BNs[c] * (double)kps / (32.0 * 32.0 * 16.0) * 32.0 / 4.0;
            const double wg_cycles = mf / 0.7 + 60.0 * (kps / BK) + 1500.0 + 8.0 * BMs[c] * BNs[c] / 64.0;
            double cost = rounds * OCC[c] * wg_cycles;
            // HBM floor of the padded problem (reads through L2 are not free either): bytes / (chip bytes per cycle)
            const double bytes = (double)tm * BMs[c] * d->K * (d->a_f32 ? 4.0 : 2.0 * d->planes) + 4.0 * (double)d->M * d->N;
            cost = cost > bytes / 2200.0 ? cost : bytes / 2200.0;
            if (splits > 1) cost += 12000.0 + (double)(splits + 1) * d->M * d->N * 8.0 / 1250.0;
            if (cost < best_cost) { best_cost = cost; best = BpPlan{c, (int)tm, (int)tn, splits, kps}; }
        }
    }
    return best;
}

template <int BM, int BN, int P, bool AF32>
int launch_nt_epi(const BpArgs& a, int epi, hipStream_t s) {
    dim3 grid(a.tiles_m * a.tiles_n, a.splits), block(NT_THREADS);
#define BP_LAUNCH(EPI_) hipLaunchKernelGGL((gemm_bp_nt_kernel<BM, BN, P, AF32, EPI_>), grid, block, 0, s, a)
    switch (epi) {
        case BEPI_PLAIN: BP_LAUNCH(BEPI_PLAIN); break;
        case BEPI_GELU: BP_LAUNCH(BEPI_GELU); break;
        case BEPI_DGELU: BP_LAUNCH(BEPI_DGELU); break;
        case BEPI_FULL: BP_LAUNCH(BEPI_FULL); break;
        case BEPI_DGELU_RC: if constexpr (!AF32) { BP_LAUNCH(BEPI_DGELU_RC); break; } else return 1;
        default: return 1;
    }
#undef BP_LAUNCH
    return 0;
}

template <int BM, int BN>
int launch_nt(const BpArgs& a, int planes, int epi, hipStream_t s) {
    if (planes == 2) return a.a_f32 ? launch_nt_epi<BM, BN, 2, true>(a, epi, s) : launch_nt_epi<BM, BN, 2, false>(a, epi, s);
    return a.a_f32 ? launch_nt_epi<BM, BN, 1, true>(a, epi, s) : launch_nt_epi<BM, BN, 1, false>(a, epi, s);
}

}  // namespace

int mdvit_gemm_splitk_reduce(const float* slab, const float* bias, float* C, long ldc, int M, int N, int splits, int accumulate, hipStream_t s);

extern "C" size_t mdvit_gemm_planes_ws_bytes(const MdvitPlaneGemmDesc* d) {
    if (d == nullptr || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->trans) return 0;
    const BpPlan pl = plan_bp(d);
    return pl.splits > 1 ? sizeof(float) * (size_t)pl.splits * d->M * d->N : 0;
}

extern "C" int mdvit_gemm_planes_force_plan(int32_t cfg, int32_t splits) {
    g_bp_force_cfg = (cfg >= 0 && cfg <= 7) ? cfg : -1;
    g_bp_force_splits = splits > 0 ? splits : 0;
    return MDVIT_OK;
}

extern "C" int mdvit_gemm_planes_plan(const MdvitPlaneGemmDesc* d, int32_t* tile_m, int32_t* tile_n, int32_t* splits) {
    MDVIT_CHECK_ARG(d != nullptr && d->M > 0 && d->N > 0 && d->K > 0, MDVIT_E_SHAPE, "gemm_planes_plan: bad descriptor");
    const BpPlan pl = plan_bp(d);
    if (tile_m) *tile_m = pl.cfg >= 6 ? 128 : (pl.cfg >= 3 ? 256 : (pl.cfg == 2 ? 64 : 128));
    if (tile_n) *tile_n = pl.cfg == 6 ? 160 : (pl.cfg == 7 ? 128 : (pl.cfg >= 3 ? 256 : (pl.cfg == 0 ? 128 : 64)));
    if (splits) *splits = pl.splits;
    return MDVIT_OK;
}

extern "C" int mdvit_gemm_planes(const MdvitPlaneGemmDesc* d, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(d != nullptr, MDVIT_E_SHAPE, "gemm_planes: null descriptor");
    MDVIT_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, MDVIT_E_SHAPE, "gemm_planes: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    MDVIT_CHECK_ARG(d->A && d->B && (d->C || d->Cp), MDVIT_E_SHAPE, "gemm_planes: null operand");
    MDVIT_CHECK_ARG(d->planes == 1 || d->planes == 2, MDVIT_E_SHAPE, "gemm_planes: planes must be 1 (bf16) or 2 (bf16x3), got %d", d->planes);
    MDVIT_CHECK_ARG(!d->trans, MDVIT_E_SHAPE, "gemm_planes: the TN (weight gradient) layout goes through mdvit_gemm_planes_tn");
    MDVIT_CHECK_ARG(d->K % BK == 0, MDVIT_E_SHAPE, "gemm_planes: K must be a multiple of %d (K=%d)", BK, d->K);
    MDVIT_CHECK_ARG(d->N % 4 == 0, MDVIT_E_SHAPE, "gemm_planes: N must be a multiple of 4 (N=%d)", d->N);
    MDVIT_CHECK_ARG(aligned16(d->A) && aligned16(d->B) && (d->lda % 8 == 0) && (d->ldb % 8 == 0) && (d->a_plane % 8 == 0) && (d->b_plane % 8 == 0),
                    MDVIT_E_ALIGN, "gemm_planes: operands must be 16-byte aligned, leading dimensions / plane strides %% 8 == 0 (lda=%ld ldb=%ld)", (long)d->lda, (long)d->ldb);
    MDVIT_CHECK_ARG((!d->C || (aligned16(d->C) && d->ldc % 4 == 0)) && (!d->Cp || ((reinterpret_cast<uintptr_t>(d->Cp) & 7) == 0 && d->ldcp % 4 == 0 && d->c_plane % 4 == 0)) &&
                    (!d->bias || aligned16(d->bias)) && (!d->residual || (aligned16(d->residual) && d->ldr % 4 == 0)) &&
                    (!d->gelu_u || (aligned16(d->gelu_u) && d->ldu % 4 == 0)) && (!d->U || (aligned16(d->U) && d->ldu_out % 4 == 0)),
                    MDVIT_E_ALIGN, "gemm_planes: outputs / epilogue operands must be 16-byte aligned with leading dimensions %% 4 == 0");
    MDVIT_CHECK_ARG(d->epi != MDVIT_EPI_DGELU || d->gelu_u || d->rc_a, MDVIT_E_SHAPE, "gemm_planes: DGELU needs gelu_u (or rc_a / rc_b to recompute it)");
    if (d->rc_a) {
        MDVIT_CHECK_ARG(d->epi == MDVIT_EPI_DGELU && !d->gelu_u && d->rc_b && d->rc_k > 0 && d->rc_k % BK == 0 && !d->a_f32, MDVIT_E_SHAPE,
                        "gemm_planes: rc_a / rc_b (recomputed pre-activation) go with the DGELU epilogue, plane operands, rc_k %% 32 == 0 and no gelu_u");
        MDVIT_CHECK_ARG(aligned16(d->rc_a) && aligned16(d->rc_b) && d->rc_lda % 8 == 0 && d->rc_ldb % 8 == 0 && (!d->rc_bias || aligned16(d->rc_bias)),
                        MDVIT_E_ALIGN, "gemm_planes: rc operands must be 16-byte aligned, leading dimensions %% 8 == 0");
    }
    MDVIT_CHECK_ARG(!(d->e_drop_p > 0.f) || (long)d->M * d->N < (1L << 32), MDVIT_E_SHAPE, "gemm_planes: dropout index space exceeds 2^32");
    MDVIT_CHECK_ARG(!d->accumulate || (d->C && !d->Cp), MDVIT_E_SHAPE, "gemm_planes: accumulate needs the fp32 output only");

    BpArgs a;
    memset(&a, 0, sizeof(a));
    a.A = d->A; a.lda = d->lda; a.a_plane = d->a_plane; a.a_f32 = d->a_f32;
    a.B = (const uint16_t*)d->B; a.ldb = d->ldb; a.b_plane = d->b_plane;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.C = d->C; a.ldc = d->ldc; a.Cp = (uint16_t*)d->Cp; a.ldcp = d->ldcp; a.c_plane = d->c_plane;
    a.U = d->U; a.ldu_out = d->ldu_out;
    a.bias = d->bias;
    a.e_drop = d->e_drop_p > 0.f; a.e_k0 = d->e_key0; a.e_k1 = d->e_key1;
    a.e_thresh = mdvit_drop_thresh(d->e_drop_p); a.e_inv_keep = 1.f / (1.f - d->e_drop_p);
    a.e_rowscale = d->e_rowscale; a.e_rows_per_scale = d->e_rows_per_scale > 0 ? d->e_rows_per_scale : 1;
    a.residual = d->residual; a.ldr = d->ldr; a.gelu_u = d->gelu_u; a.ldu = d->ldu;
    a.rc_a = (const uint16_t*)d->rc_a; a.rc_lda = d->rc_lda; a.rc_a_plane = d->rc_a_plane;
    a.rc_b = (const uint16_t*)d->rc_b; a.rc_ldb = d->rc_ldb; a.rc_b_plane = d->rc_b_plane; a.rc_bias = d->rc_bias; a.rc_k = d->rc_k;
    a.accumulate = d->accumulate;
    a.seed = d->drop_seed;

    int epi = BEPI_PLAIN;
    if (d->epi == MDVIT_EPI_GELU_DUAL) epi = BEPI_GELU;
    else if (d->epi == MDVIT_EPI_DGELU) epi = d->rc_a ? BEPI_DGELU_RC : BEPI_DGELU;
    else if (a.e_drop || d->e_rowscale || d->residual) epi = BEPI_FULL;
    MDVIT_CHECK_ARG(epi == BEPI_PLAIN || !d->accumulate, MDVIT_E_SHAPE, "gemm_planes: accumulate is only defined for the plain epilogue");

    const BpPlan pl = plan_bp(d);
    a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n; a.splits = pl.splits; a.k_per_split = pl.kps;
    if (pl.splits > 1) {
        const size_t need = sizeof(float) * (size_t)pl.splits * d->M * d->N;
        MDVIT_CHECK_ARG(d->ws != nullptr && d->ws_bytes >= need, MDVIT_E_WORKSPACE,
                        "gemm_planes: split reduction needs %zu bytes of workspace (mdvit_gemm_planes_ws_bytes), got %zu", need, (size_t)d->ws_bytes);
        a.slab = (float*)d->ws;
    }
    int rc;
    if (pl.cfg >= 6) {
        MDVIT_CHECK_ARG(mdvit_gemm_pm_ok(a, pl.cfg, d->planes, epi), MDVIT_E_SHAPE,
                        "gemm_planes: the 128-row phase-split kernel needs fp32 A, two weight planes, one K range and K %% 32 == 0 (K=%d planes=%d a_f32=%d)", d->K, d->planes, d->a_f32);
        rc = mdvit_gemm_pm_launch(a, pl.cfg, epi, s);
    }
    else if (pl.cfg >= 3) {
        MDVIT_CHECK_ARG(mdvit_gemm_ph_ok(a, pl.cfg, d->planes, epi, pl.kps), MDVIT_E_SHAPE,
                        "gemm_planes: the 256-wide phase-split kernel needs plane operands, K (per split) a multiple of %d and >= %d (K=%d, per split %d)",
                        d->planes == 2 ? 32 : 64, d->planes == 2 ? 64 : 128, d->K, pl.kps);
        rc = mdvit_gemm_ph_launch(a, pl.cfg, d->planes, epi, s);
    }
    else if (pl.cfg == 0) rc = launch_nt<128, 128>(a, d->planes, epi, s);
    else if (pl.cfg == 1) rc = launch_nt<128, 64>(a, d->planes, epi, s);
    else rc = launch_nt<64, 64>(a, d->planes, epi, s);
    MDVIT_CHECK_ARG(rc == 0, MDVIT_E_SHAPE, "gemm_planes: this (a_f32=%d, epilogue=%d) combination is not built", d->a_f32, epi);
    if (pl.splits > 1) {
        const int rr = mdvit_gemm_splitk_reduce(a.slab, d->bias, d->C, (long)d->ldc, d->M, d->N, pl.splits, d->accumulate, s);
        if (rr != MDVIT_OK) return rr;
    }
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_split_planes_many(const void* items_dev, int32_t n, int32_t blocks_per_item, int32_t planes, void* stream) {
    MDVIT_CHECK_ARG(items_dev && n > 0 && blocks_per_item > 0 && (planes == 1 || planes == 2), MDVIT_E_SHAPE, "split_planes_many: bad arguments");
    hipLaunchKernelGGL(split_planes_many_kernel, dim3(blocks_per_item, n), dim3(256), 0, (hipStream_t)stream, (const long long*)items_dev, planes);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_split_planes_t(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t plane_stride, int32_t rows, int32_t cols, int32_t transpose,
                                    int32_t planes, void* stream) {
    MDVIT_CHECK_ARG(in && out && rows > 0 && cols > 0 && (planes == 1 || planes == 2), MDVIT_E_SHAPE, "split_planes_t: bad arguments (rows=%d cols=%d)", rows, cols);
    const int tiles = cdiv(rows, 32) * cdiv(cols, 32);
    hipLaunchKernelGGL(split_planes_t_kernel, dim3(tiles < 1024 ? tiles : 1024), dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, (uint16_t*)out, (long)ld_out,
                       (long)plane_stride, rows, cols, transpose, planes);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_split_planes(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t plane_stride, int64_t rows, int32_t cols, int32_t planes,
                                  void* stream) {
    MDVIT_CHECK_ARG(in && out && rows > 0 && cols > 0 && cols % 8 == 0 && ld_in % 4 == 0 && ld_out % 8 == 0 && plane_stride % 8 == 0 && (planes == 1 || planes == 2),
                    MDVIT_E_SHAPE, "split_planes: need cols %% 8 == 0, ld_in %% 4 == 0, ld_out / plane stride %% 8 == 0 (rows=%ld cols=%d)", (long)rows, cols);
    MDVIT_CHECK_ARG(aligned16(in) && aligned16(out), MDVIT_E_ALIGN, "split_planes: operands must be 16-byte aligned");
    const long total = rows * (cols / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(split_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, (uint16_t*)out, (long)ld_out, (long)plane_stride,
                       (long)rows, cols / 8, planes);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
