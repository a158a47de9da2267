// softmax(Q K^T / sqrt(d)) V on the fp32 matrix cores for the DeiT trunk of TransFuse_S_adapt (Attention_Sup, vision_transformer.py:143-150
// with the Domain Adapter's per-(sample, channel) scale of :151-169 applied to the output): N = 256 tokens, head dimension 64.
// One workgroup = one (sample, head); its two [256, 64] operand matrices sit in LDS TRANSPOSED ([d][token], 257-float rows), which
// serves both MFMA roles without a bank conflict: as an A operand (lanes along tokens: consecutive addresses) and as a B operand
// (lanes along d: stride 257 = 1 mod 32).  v_mfma_f32_32x32x2_f32 throughout (exact fp32 products, fp32 accumulation).
//
// The score tile is computed TRANSPOSED (S^T = K Q^T): the MFMA result then holds, per lane, ONE query column and 16 keys per tile
// in registers, so the softmax over keys is an in-lane reduction plus one exchange with lane ^ 32, and the probabilities go straight
// back into the next MFMA as its A operand (row = query = lane & 31, k = key): lanes 0-31 feed the key of register r, lanes 32-63
// the key 4 rows further down -- the contraction order is free as long as the B operand (V rows) follows the same pairing.  Nothing
// of size [N, N] ever leaves the registers: the forward keeps only the row log-sum-exp, the backward recomputes the probabilities.
//   forward : S^T tiles (8 x 32 MFMAs per 32 queries) -> softmax -> O = P V (256 MFMAs)            out = a * O, lse
//   backward: prep   delta[q] = sum_d g out (= rowsum(dO * O)),  e[c] = sum_n g out (the adapter's gradient carrier)
//             rows   per 32 queries and key tile: S^T, dP^T = V dO^T, dS = P (dP - delta) / 8  ->  dQ += dS K
//             keys   per 32 keys and query tile:  S = Q K^T, dP = dO V^T, P, dS                 ->  dV += P^T dO, dK += dS^T Q
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int SD = 64;            // head dimension
constexpr int SN = 256;           // tokens
constexpr int SLD = 257;          // LDS row length of a transposed [64][256] operand
constexpr int SMEM_BYTES = 2 * SD * SLD * 4;

// stage X[token][0..63] (row stride ld floats; optionally times scale[0..63]) as Xt[d][token]; 4 wavefronts x 64 rows each
__device__ __forceinline__ void stage_t(float* __restrict__ Xt, const float* __restrict__ X, long ld, const float* __restrict__ scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float sc = scale ? scale[lane] : 1.f;
#pragma unroll 4
    for (int i = 0; i < 64; i += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = X[(long)(wave * 64 + i + j) * ld + lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) Xt[lane * SLD + wave * 64 + i + j] = v[j] * sc;
    }
}

// 32 contiguous floats of one row -> 32 registers (the lane's half of the head dimension)
__device__ __forceinline__ void load_half_row(float (&f)[32], const float* __restrict__ row) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(row + 4 * i);
        f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
    }
}

// row index inside a 32x32 MFMA result tile held by register r of this lane
__device__ __forceinline__ int drow(int r, int lhi) { return (r & 3) + 8 * (r >> 2) + 4 * lhi; }

__global__ __launch_bounds__(256) void sdpa_mfma_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ a, float* __restrict__ out,
                                                            float* __restrict__ lse, int C, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Kt = smem; float* Vt = smem + SD * SLD;
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const long ld = 3L * C;
    const float* base = qkv + (long)b * SN * ld + h * SD;
    stage_t(Kt, base + C, ld, nullptr);
    stage_t(Vt, base + 2 * C, ld, nullptr);
    __syncthreads();
    for (int qt = wave; qt < SN / 32; qt += 4) {
        const int q = qt * 32 + l31;
        float qf[32];
        load_half_row(qf, base + (long)q * ld + 32 * lhi);
        f32x16 acc[8];
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {            // one tile at a time: 8 interleaved accumulator chains spill
            const float* kcol = Kt + 32 * lhi * SLD + kt * 32 + l31;
#pragma unroll
            for (int s = 0; s < 32; ++s) acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kcol[s * SLD], qf[s], acc[kt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // softmax over the keys of this lane's query: 128 values here, the other 128 in lane ^ 32
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[kt][r] *= scale; m = fmaxf(m, acc[kt][r]); }
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[kt][r] = __expf(acc[kt][r] - m); sum += acc[kt][r]; }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;
        if (lhi == 0) lse[((long)b * heads + h) * SN + q] = m + __logf(sum);
        // O = P V: A = probabilities from the registers, B = V rows of the same key pairing
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = acc[kt][r] * inv;
                const int key = kt * 32 + drow(r, lhi);
                o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, Vt[l31 * SLD + key], o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, Vt[(32 + l31) * SLD + key], o[1], 0, 0, 0);
            }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int c = h * SD + dt * 32 + l31;
            const float av = a ? a[(long)b * C + c] : 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((long)b * SN + qt * 32 + drow(r, lhi)) * C + c] = o[dt][r] * av;
        }
    }
}

// delta[b,h,n] = sum_d g out over the head's channels; e_part[b][j][c] = sum over the j-th quarter of the tokens of g out (the rows kernel
// adds the four quarters in order: e[b,c] = sum_n g out, the adapter's gradient carrier).  Workgroup = (sample, token quarter), 4 wavefronts.
// Round 6: the products are formed with lane = channel (coalesced 256-byte head rows; the per-channel e sums need no exchange), parked in LDS [token][65] and the
// per-token sums are taken with lane = TOKEN (a conflict-free walk of its row) -- the first form reduced every (token, head) over the wave with six dependent
// ds_bpermute rounds, 16 tokens x 6 heads in series per wave: 76 us for 25 MB on 128 workgroups (8 launches per TransFuse step on the DeiT branch's chain).
__global__ __launch_bounds__(256) void sdpa_prep_kernel(const float* __restrict__ g, const float* __restrict__ out, float* __restrict__ delta,
                                                        float* __restrict__ e_part, int C, int heads) {
    __shared__ float s_e[4][6 * 64];
    __shared__ float s_p[64][65];          // one head at a time: [token of the quarter][channel], 65-float rows
    const int b = blockIdx.x, j = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int TQ = SN / 4;             // 64 tokens per workgroup
    float ecol[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < heads; ++h) {
        float pv[TQ / 4];
#pragma unroll
        for (int i = 0; i < TQ / 4; ++i) {          // this wave's 16 tokens of the quarter: all loads in flight
            const long row = ((long)b * SN + j * TQ + wave + 4 * i) * C + h * SD + lane;
            pv[i] = g[row] * out[row];
        }
        float ec = 0.f;
#pragma unroll
        for (int i = 0; i < TQ / 4; ++i) { ec += pv[i]; s_p[wave + 4 * i][lane] = pv[i]; }          // (tokens in increasing order per wave, as before)
        ecol[h] = ec;
        __syncthreads();
        if (threadIdx.x < TQ) {                     // lane = token: the 64 channels of its row, in channel order
            float d = 0.f;
#pragma unroll 16
            for (int c = 0; c < SD; ++c) d += s_p[threadIdx.x][c];
            delta[((long)b * heads + h) * SN + j * TQ + threadIdx.x] = d;
        }
        __syncthreads();
    }
#pragma unroll
    for (int h = 0; h < 6; ++h) s_e[wave][h * 64 + lane] = ecol[h];
    __syncthreads();
    for (int c = threadIdx.x; c < heads * SD; c += 256)
        e_part[((long)b * 4 + j) * C + c] = (s_e[0][c] + s_e[1][c]) + (s_e[2][c] + s_e[3][c]);
}

__global__ __launch_bounds__(256) void sdpa_mfma_bwd_rows_kernel(const float* __restrict__ g, const float* __restrict__ qkv, const float* __restrict__ lse,
                                                                 const float* __restrict__ delta, const float* __restrict__ e_part, const float* __restrict__ a,
                                                                 float* __restrict__ dqkv, float* __restrict__ e, int C, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Kt = smem; float* Vt = smem + SD * SLD;
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const long ld = 3L * C;
    const float* base = qkv + (long)b * SN * ld + h * SD;
    stage_t(Kt, base + C, ld, nullptr);
    stage_t(Vt, base + 2 * C, ld, nullptr);
    if (e && threadIdx.x < SD) {            // this head's 64 channels of e: the four token quarters of the prep pass, in order
        const int c = h * SD + threadIdx.x;
        const float* ep = e_part + (long)b * 4 * C + c;
        e[(long)b * C + c] = (ep[0] + ep[C]) + (ep[2 * C] + ep[3 * C]);
    }
    __syncthreads();
    for (int qt = wave; qt < SN / 32; qt += 4) {
        const int q = qt * 32 + l31;
        float qf[32], df[32];
        load_half_row(qf, base + (long)q * ld + 32 * lhi);
        load_half_row(df, g + ((long)b * SN + q) * C + h * SD + 32 * lhi);
        if (a) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 av = *reinterpret_cast<const float4*>(a + (long)b * C + h * SD + 32 * lhi + 4 * i);
                df[4 * i] *= av.x; df[4 * i + 1] *= av.y; df[4 * i + 2] *= av.z; df[4 * i + 3] *= av.w;
            }
        }
        const float l = lse[((long)b * heads + h) * SN + q], dl = delta[((long)b * heads + h) * SN + q];
        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll 1
        for (int kt = 0; kt < 8; ++kt) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                st = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[(32 * lhi + s) * SLD + kt * 32 + l31], qf[s], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vt[(32 * lhi + s) * SLD + kt * 32 + l31], df[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(st[r] * scale - l);
                const float ds = p * (dp[r] - dl) * scale;
                const int key = kt * 32 + drow(r, lhi);
                dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, Kt[l31 * SLD + key], dq[0], 0, 0, 0);
                dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, Kt[(32 + l31) * SLD + key], dq[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqkv[((long)b * SN + qt * 32 + drow(r, lhi)) * ld + h * SD + dt * 32 + l31] = dq[dt][r];
    }
}

__global__ __launch_bounds__(256) void sdpa_mfma_bwd_keys_kernel(const float* __restrict__ g, const float* __restrict__ qkv, const float* __restrict__ lse,
                                                                 const float* __restrict__ delta, const float* __restrict__ a, float* __restrict__ dqkv,
                                                                 int C, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qt = smem; float* Gt = smem + SD * SLD;          // Gt = (g * a)^T
    __shared__ float s_l[SN], s_d[SN];
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const long ld = 3L * C;
    const float* base = qkv + (long)b * SN * ld + h * SD;
    stage_t(Qt, base, ld, nullptr);
    stage_t(Gt, g + (long)b * SN * C + h * SD, (long)C, a ? a + (long)b * C + h * SD : nullptr);
    s_l[threadIdx.x] = lse[((long)b * heads + h) * SN + threadIdx.x];
    s_d[threadIdx.x] = delta[((long)b * heads + h) * SN + threadIdx.x];
    __syncthreads();
    for (int kt = wave; kt < SN / 32; kt += 4) {
        const int key = kt * 32 + l31;
        float kf[32], vf[32];
        load_half_row(kf, base + C + (long)key * ld + 32 * lhi);
        load_half_row(vf, base + 2 * C + (long)key * ld + 32 * lhi);
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
#pragma unroll 1
        for (int qt = 0; qt < 8; ++qt) {
            f32x16 sc, dp;               // [row = query][col = this lane's key]
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Qt[(32 * lhi + s) * SLD + qt * 32 + l31], kf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Gt[(32 * lhi + s) * SLD + qt * 32 + l31], vf[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qt * 32 + drow(r, lhi);
                const float p = __expf(sc[r] * scale - s_l[q]);
                const float ds = p * (dp[r] - s_d[q]) * scale;
                // A = P^T / dS^T (row = key = lane & 31, k = query); B = dO / Q rows of the same query pairing
                dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, Gt[l31 * SLD + q], dv[0], 0, 0, 0);
                dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p, Gt[(32 + l31) * SLD + q], dv[1], 0, 0, 0);
                dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, Qt[l31 * SLD + q], dk[0], 0, 0, 0);
                dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds, Qt[(32 + l31) * SLD + q], dk[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long o = ((long)b * SN + kt * 32 + drow(r, lhi)) * ld + h * SD + dt * 32 + l31;
                dqkv[o + C] = dk[dt][r];
                dqkv[o + 2 * C] = dv[dt][r];
            }
    }
}

int set_lds(const void* k, int& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 31) dev = 0;
    if (mask & (1 << dev)) return MDVIT_OK;
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    if (e != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "sdpa: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    mask |= 1 << dev;
    return MDVIT_OK;
}

}  // namespace

extern "C" int mdvit_sdpa_mfma_fwd(const float* qkv, const float* a, float* out, float* lse, int32_t B, int32_t N, int32_t C, int32_t heads, void* stream) {
    MDVIT_CHECK_ARG(qkv && out && lse && B > 0 && N == SN && heads > 0 && heads <= 6 && C == heads * SD, MDVIT_E_SHAPE,
                    "sdpa_mfma_fwd: built for N == 256, head dimension 64, <= 6 heads (N=%d C=%d heads=%d)", N, C, heads);
    MDVIT_CHECK_ARG(aligned16(qkv) && (!a || aligned16(a)), MDVIT_E_ALIGN, "sdpa_mfma_fwd: operands must be 16-byte aligned");
    static int mask = 0;
    const int rc = set_lds(reinterpret_cast<const void*>(&sdpa_mfma_fwd_kernel), mask);
    if (rc != MDVIT_OK) return rc;
    hipLaunchKernelGGL(sdpa_mfma_fwd_kernel, dim3(heads, B), dim3(256), SMEM_BYTES, (hipStream_t)stream, qkv, a, out, lse, C, heads, 0.125f);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

/* delta: scratch [2][B, heads, N] floats (row sums, then the partial column sums of the prep pass) */
extern "C" int mdvit_sdpa_mfma_bwd(const float* g, const float* qkv, const float* lse, const float* out, const float* a, float* dqkv, float* e, float* delta,
                                   int32_t B, int32_t N, int32_t C, int32_t heads, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(g && qkv && lse && out && dqkv && delta && B > 0 && N == SN && heads > 0 && heads <= 6 && C == heads * SD, MDVIT_E_SHAPE,
                    "sdpa_mfma_bwd: built for N == 256, head dimension 64, <= 6 heads (N=%d C=%d heads=%d)", N, C, heads);
    MDVIT_CHECK_ARG((a == nullptr) == (e == nullptr), MDVIT_E_SHAPE, "sdpa_mfma_bwd: the adapter scale a and its gradient carrier e go together");
    MDVIT_CHECK_ARG(aligned16(g) && aligned16(qkv) && (!a || aligned16(a)), MDVIT_E_ALIGN, "sdpa_mfma_bwd: operands must be 16-byte aligned");
    static int m1 = 0, m2 = 0;
    int rc = set_lds(reinterpret_cast<const void*>(&sdpa_mfma_bwd_rows_kernel), m1);
    if (rc == MDVIT_OK) rc = set_lds(reinterpret_cast<const void*>(&sdpa_mfma_bwd_keys_kernel), m2);
    if (rc != MDVIT_OK) return rc;
    float* e_part = delta + (long)B * heads * N;
    hipLaunchKernelGGL(sdpa_prep_kernel, dim3(B, 4), dim3(256), 0, s, g, out, delta, e_part, C, heads);
    hipLaunchKernelGGL(sdpa_mfma_bwd_rows_kernel, dim3(heads, B), dim3(256), SMEM_BYTES, s, g, qkv, lse, delta, e_part, a, dqkv, e, C, heads, 0.125f);
    hipLaunchKernelGGL(sdpa_mfma_bwd_keys_kernel, dim3(heads, B), dim3(256), SMEM_BYTES, s, g, qkv, lse, delta, a, dqkv, C, heads, 0.125f);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
