// Fused MLP of the C = 64 stages, second generation (Mlp.forward mpvit.py:71-78 inside SerialBlock, mdvit.py:357-360, and its
// backward data path).  Why: ablating the first-generation kernels (mlp.hip; tools/mlp_check.py) showed that neither HBM nor the matrix
// cores nor the GELU bound them -- with the h store, the weight reloads, the GELU and the second product ALL switched off the forward
// still took 52 % of its time: three workgroup barriers per 64-wide hidden chunk, the hi/lo split of both weight chunks by every
// workgroup, and the round trip of the hidden activations through LDS between the two products serialise four waves per workgroup.
// Here a WAVE owns 32 tokens end to end and the waves of a workgroup share nothing but the weight chunks:
//   * weights arrive pre-split (the per-step bf16 planes of ops.refresh_weight_planes) and go L2 -> LDS by global_load_lds into a
//     two-deep ring: no conversion work, ONE barrier per hidden chunk;
//   * the x (and, backward, gm) MFMA fragments of the wave's 32 tokens live in registers for the whole walk;
//   * the hidden chunk never leaves the register file: product 1 runs as D[hidden][token], so a lane holds 16 hidden values of ITS
//     token -- which is the operand layout of product 2 with the contraction index permuted ((i&3) + 8(i>>2) + 4*lhi within a
//     16-block; the weight fragments of product 2 are read in the same order: two ds_read_b64 per fragment).
// bf16x3 arithmetic as everywhere (hi*lo + lo*hi + hi*hi per 16-deep step, k ascending); the permuted contraction order changes the
// summation order INSIDE one MFMA only: results equal the GEMM path to fp32 round-off (tests: 2e-6), not bit for bit.
#include "common.h"

typedef float m2_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 m2_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned m2_u4 __attribute__((ext_vector_type(4)));
typedef unsigned m2_u2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int C = 64, HC = 64;                 // channels, hidden chunk
constexpr int WTILE = 64 * 128;                // bytes of one [64 rows][64 k] bf16 plane in LDS (128-byte rows)

struct Mlp2Args {
    const float* x; const float* gm; const float* res; const float* rowscale; const float* b1; const float* b2;
    const uint16_t* W1p;    // planes of W1  [P][Hd][C]   (rows = hidden, k = c)        product "u = x W1^T"
    const uint16_t* W2p;    // planes of W2  [P][C][Hd]   (rows = c_out, k = hidden)    forward product 2
    const uint16_t* W2tp;   // planes of W2^T [P][Hd][C]  (rows = hidden, k = c)        backward "d = gm W2"
    const uint16_t* W1tp;   // planes of W1^T [P][C][Hd]  (rows = c, k = hidden)        backward "dx = d W1"
    float* h; float* y; float* du; float* dx;
    int M, Hd, rows_per_scale;
    int drop; uint32_t k1a, k1b, k2a, k2b, thresh; float inv_keep;
    const uint32_t* seed;
};

// one [64 rows][64 k] plane: 8 pieces of 1 KiB (8 rows x 128 B); lane i -> row 8q + (i >> 3), physical 16-byte chunk i & 7,
// fetching logical chunk (i & 7) ^ ((row >> 1) & 7)  [conflict-free ds_read_b128 / b64 of row-per-lane fragments on 128-byte rows]
__device__ __forceinline__ void m2_glds_piece(const uint16_t* __restrict__ src, long ld, int q, int lane, char* dst) {
    const int row = q * 8 + (lane >> 3);
    const int f = (row >> 1) & 7;
    __builtin_amdgcn_global_load_lds(src + (long)row * ld + (((lane & 7) ^ f) << 3), dst + q * 1024, 16, 0, 0);
}

__device__ __forceinline__ m2_bf16x8 m2_frag128(const char* tile, int row, int kb, int lhi) {       // k = 16 kb + 8 lhi .. +7, natural order
    const int c = 2 * kb + lhi;
    return __builtin_bit_cast(m2_bf16x8, *reinterpret_cast<const m2_u4*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)));
}
// permuted order of a register-chained operand: slots i = 0..7 <-> k = 16 kb + (i & 3) + 8 (i >> 2) + 4 lhi
__device__ __forceinline__ m2_bf16x8 m2_frag_perm(const char* tile, int row, int kb, int lhi) {
    const int f = (row >> 1) & 7;
    const int c0 = 2 * kb, c1 = 2 * kb + 1;                   // 16-byte chunks holding k 16kb..+7 and 16kb+8..+15; the lane wants 8 bytes at 8*lhi of each
    const m2_u2 a = *reinterpret_cast<const m2_u2*>(tile + row * 128 + ((c0 ^ f) << 4) + 8 * lhi);
    const m2_u2 b = *reinterpret_cast<const m2_u2*>(tile + row * 128 + ((c1 ^ f) << 4) + 8 * lhi);
    const m2_u4 v = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(m2_bf16x8, v);
}

__device__ __forceinline__ void m2_split8(const float* v, m2_bf16x8& hi, m2_bf16x8& lo) {
    uint2 h0, l0, h1, l1;
    mdvit_split_bf16x3(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
    mdvit_split_bf16x3(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
    const m2_u4 hv = {h0.x, h0.y, h1.x, h1.y}, lv = {l0.x, l0.y, l1.x, l1.y};
    hi = __builtin_bit_cast(m2_bf16x8, hv);
    lo = __builtin_bit_cast(m2_bf16x8, lv);
}

#define M2_MFMA3(acc, bh, bl, ah, al)                                           \
    do {                                                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc, 0, 0, 0);    \
    } while (0)

// the x / gm fragments of a wave's 32 tokens: lane (token l31, lhi) reads k = 16 kb + 8 lhi .. +7 of its row for kb = 0..3
__device__ __forceinline__ void m2_load_rows(const float* __restrict__ src, int row, int M, int lhi, m2_bf16x8 (&hi)[4], m2_bf16x8 (&lo)[4]) {
    float v[4][8];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (row < M) {
            a = *reinterpret_cast<const float4*>(src + (long)row * C + 16 * kb + 8 * lhi);
            b = *reinterpret_cast<const float4*>(src + (long)row * C + 16 * kb + 8 * lhi + 4);
        }
        v[kb][0] = a.x; v[kb][1] = a.y; v[kb][2] = a.z; v[kb][3] = a.w; v[kb][4] = b.x; v[kb][5] = b.y; v[kb][6] = b.z; v[kb][7] = b.w;
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) m2_split8(v[kb], hi[kb], lo[kb]);
}

// ---- forward:  h = drop1(gelu(x W1^T + b1)) (written: the weight gradients' operand),  y = res + rowscale * drop2(h W2^T + b2)
template <int NW>
__global__ __launch_bounds__(NW * 64) void mlp2_fwd_kernel(Mlp2Args p) {
    constexpr int STAGE = 4 * WTILE;                      // W1c hi, lo | W2c hi, lo
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int row = blockIdx.x * (NW * 32) + wave * 32 + l31;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1, k2a = p.k2a ^ s0, k2b = p.k2b + s1;
    const long w_plane = (long)p.Hd * C;
    auto issue = [&](int hc0, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int q0 = 0; q0 < 32 / NW; ++q0) {            // 32 pieces: tensor t = q / 8 (W1 hi, W1 lo, W2 hi, W2 lo), piece q % 8
            const int q = q0 * NW + wave, t = q >> 3, pq = q & 7;
            if (t < 2) m2_glds_piece(p.W1p + t * w_plane + (long)hc0 * C, C, pq, lane, base + t * WTILE);
            else m2_glds_piece(p.W2p + (t - 2) * w_plane + hc0, p.Hd, pq, lane, base + t * WTILE);
        }
    };
    issue(0, 0);
    m2_bf16x8 xh[4], xl[4];
    m2_load_rows(p.x, row, p.M, lhi, xh, xl);
    m2_f32x16 yacc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[cb][r] = 0.f;

    int buf = 0;
    for (int hc0 = 0; hc0 < p.Hd; hc0 += HC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // this chunk's weights are in `buf`; everybody is done with the other buffer
        if (hc0 + HC < p.Hd) issue(hc0 + HC, buf ^ 1);
        const char* W1hi = smem + buf * STAGE; const char* W1lo = W1hi + WTILE; const char* W2hi = W1lo + WTILE; const char* W2lo = W2hi + WTILE;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            // product 1: u[hidden 32nb + ..][token] over k = c
            m2_f32x16 u;
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const m2_bf16x8 bh = m2_frag128(W1hi, nb * 32 + l31, kb, lhi), bl = m2_frag128(W1lo, nb * 32 + l31, kb, lhi);
                M2_MFMA3(u, bh, bl, xh[kb], xl[kb]);
            }
            // bias + GELU + dropout in registers; h to HBM; the same registers are product 2's operand
            float hv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hd = hc0 + nb * 32 + 8 * q + 4 * lhi;
                const float4 b4 = *reinterpret_cast<const float4*>(p.b1 + hd);
                float4 v = make_float4(gelu_f(u[4 * q + 0] + b4.x), gelu_f(u[4 * q + 1] + b4.y), gelu_f(u[4 * q + 2] + b4.z), gelu_f(u[4 * q + 3] + b4.w));
                if (p.drop) {
                    const float4 ds = mdvit_drop_scale4(k1a, k1b, (uint32_t)((long)row * p.Hd + hd), p.thresh, p.inv_keep);
                    v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
                }
                if (p.h && row < p.M) *reinterpret_cast<float4*>(p.h + (long)row * p.Hd + hd) = v;
                hv[4 * q + 0] = v.x; hv[4 * q + 1] = v.y; hv[4 * q + 2] = v.z; hv[4 * q + 3] = v.w;
            }
            // product 2: y[c_out][token] += over the 32 hidden values just made (two 16-blocks: registers 0..7 and 8..15)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                m2_bf16x8 hh, hl;
                m2_split8(hv + 8 * half, hh, hl);
                const int kb2 = 2 * nb + half;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const m2_bf16x8 bh = m2_frag_perm(W2hi, cb * 32 + l31, kb2, lhi), bl = m2_frag_perm(W2lo, cb * 32 + l31, kb2, lhi);
                    M2_MFMA3(yacc[cb], bh, bl, hh, hl);
                }
            }
        }
        buf ^= 1;
    }
    if (row < p.M) {
        const float rsc = p.rowscale ? p.rowscale[row / p.rows_per_scale] : 1.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = cb * 32 + 8 * q + 4 * lhi;
                const float4 b4 = *reinterpret_cast<const float4*>(p.b2 + col);
                float4 v = make_float4(yacc[cb][4 * q + 0] + b4.x, yacc[cb][4 * q + 1] + b4.y, yacc[cb][4 * q + 2] + b4.z, yacc[cb][4 * q + 3] + b4.w);
                if (p.drop) {
                    const float4 ds = mdvit_drop_scale4(k2a, k2b, (uint32_t)((long)row * C + col), p.thresh, p.inv_keep);
                    v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
                }
                v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
                const float4 r4 = *reinterpret_cast<const float4*>(p.res + (long)row * C + col);
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                *reinterpret_cast<float4*>(p.y + (long)row * C + col) = v;
            }
    }
}

// ---- backward data path:  dx = ((gm W2) * gelu'(x W1^T + b1) * dropmask1) W1 ;  du (optional) = the bracket
template <int NW>
__global__ __launch_bounds__(NW * 64) void mlp2_bwd_kernel(Mlp2Args p) {
    constexpr int STAGE = 6 * WTILE;                      // W1c hi, lo | W2^T c hi, lo | W1^T c hi, lo
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int row = blockIdx.x * (NW * 32) + wave * 32 + l31;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1;
    const long w_plane = (long)p.Hd * C;
    auto issue = [&](int hc0, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int q0 = 0; q0 < 48 / NW; ++q0) {
            const int q = q0 * NW + wave, t = q >> 3, pq = q & 7;
            if (t < 2) m2_glds_piece(p.W1p + t * w_plane + (long)hc0 * C, C, pq, lane, base + t * WTILE);
            else if (t < 4) m2_glds_piece(p.W2tp + (t - 2) * w_plane + (long)hc0 * C, C, pq, lane, base + t * WTILE);
            else m2_glds_piece(p.W1tp + (t - 4) * w_plane + hc0, p.Hd, pq, lane, base + t * WTILE);
        }
    };
    issue(0, 0);
    m2_bf16x8 xh[4], xl[4], gh[4], gl[4];
    m2_load_rows(p.x, row, p.M, lhi, xh, xl);
    m2_load_rows(p.gm, row, p.M, lhi, gh, gl);
    m2_f32x16 dxacc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxacc[cb][r] = 0.f;

    int buf = 0;
    for (int hc0 = 0; hc0 < p.Hd; hc0 += HC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (hc0 + HC < p.Hd) issue(hc0 + HC, buf ^ 1);
        const char* base = smem + buf * STAGE;
        const char* W1hi = base; const char* W1lo = base + WTILE; const char* W2thi = base + 2 * WTILE; const char* W2tlo = base + 3 * WTILE;
        const char* W1thi = base + 4 * WTILE; const char* W1tlo = base + 5 * WTILE;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            m2_f32x16 u, d;
#pragma unroll
            for (int r = 0; r < 16; ++r) { u[r] = 0.f; d[r] = 0.f; }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const m2_bf16x8 bh = m2_frag128(W1hi, nb * 32 + l31, kb, lhi), bl = m2_frag128(W1lo, nb * 32 + l31, kb, lhi);
                M2_MFMA3(u, bh, bl, xh[kb], xl[kb]);
                const m2_bf16x8 ch = m2_frag128(W2thi, nb * 32 + l31, kb, lhi), cl = m2_frag128(W2tlo, nb * 32 + l31, kb, lhi);
                M2_MFMA3(d, ch, cl, gh[kb], gl[kb]);
            }
            float dv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hd = hc0 + nb * 32 + 8 * q + 4 * lhi;
                const float4 b4 = *reinterpret_cast<const float4*>(p.b1 + hd);
                float4 v = make_float4(d[4 * q + 0] * gelu_grad_f(u[4 * q + 0] + b4.x), d[4 * q + 1] * gelu_grad_f(u[4 * q + 1] + b4.y),
                                       d[4 * q + 2] * gelu_grad_f(u[4 * q + 2] + b4.z), d[4 * q + 3] * gelu_grad_f(u[4 * q + 3] + b4.w));
                if (p.drop) {
                    const float4 ds = mdvit_drop_scale4(k1a, k1b, (uint32_t)((long)row * p.Hd + hd), p.thresh, p.inv_keep);
                    v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
                }
                if (p.du && row < p.M) *reinterpret_cast<float4*>(p.du + (long)row * p.Hd + hd) = v;
                dv[4 * q + 0] = v.x; dv[4 * q + 1] = v.y; dv[4 * q + 2] = v.z; dv[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                m2_bf16x8 dh, dl;
                m2_split8(dv + 8 * half, dh, dl);
                const int kb2 = 2 * nb + half;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const m2_bf16x8 bh = m2_frag_perm(W1thi, cb * 32 + l31, kb2, lhi), bl = m2_frag_perm(W1tlo, cb * 32 + l31, kb2, lhi);
                    M2_MFMA3(dxacc[cb], bh, bl, dh, dl);
                }
            }
        }
        buf ^= 1;
    }
    if (row < p.M) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = cb * 32 + 8 * q + 4 * lhi;
                *reinterpret_cast<float4*>(p.dx + (long)row * C + col) = make_float4(dxacc[cb][4 * q + 0], dxacc[cb][4 * q + 1], dxacc[cb][4 * q + 2], dxacc[cb][4 * q + 3]);
            }
    }
}

int m2_set_lds(const void* k, int bytes, int& flag_dev_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 31) dev = 0;
    if (flag_dev_mask & (1 << dev)) return MDVIT_OK;
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "mlp2: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    flag_dev_mask |= 1 << dev;
    return MDVIT_OK;
}

}  // namespace

/* planes: bf16 [2][rows][cols] as written by mdvit_split_planes_t / _many (plane stride = rows * cols). */
extern "C" int mdvit_mlp2_fwd(const float* x, const void* W1p, const float* b1, const void* W2p, const float* b2, const float* res, const float* rowscale,
                              int32_t rows_per_scale, float* h, float* y, int32_t M, int32_t Cn, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1,
                              uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(Cn == 64, MDVIT_E_SHAPE, "mlp2_fwd: built for C = 64 (got %d)", Cn);
    MDVIT_CHECK_ARG(M > 0 && Hd > 0 && Hd % 64 == 0, MDVIT_E_SHAPE, "mlp2_fwd: need M > 0, hidden %% 64 == 0 (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(x && W1p && b1 && W2p && b2 && res && y, MDVIT_E_SHAPE, "mlp2_fwd: null operand");
    MDVIT_CHECK_ARG(aligned16(x) && aligned16(W1p) && aligned16(b1) && aligned16(W2p) && aligned16(b2) && aligned16(res) && aligned16(y) && (!h || aligned16(h)),
                    MDVIT_E_ALIGN, "mlp2_fwd: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp2_fwd: dropout index space exceeds 2^32");
    Mlp2Args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.W1p = (const uint16_t*)W1p; a.b1 = b1; a.W2p = (const uint16_t*)W2p; a.b2 = b2; a.res = res; a.rowscale = rowscale; a.h = h; a.y = y;
    a.M = M; a.Hd = Hd; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    a.drop = drop_p > 0.f; a.k1a = key1_0; a.k1b = key1_1; a.k2a = key2_0; a.k2b = key2_1;
    a.thresh = (uint32_t)((double)drop_p * 4294967296.0); a.inv_keep = 1.f / (1.f - drop_p);
    a.seed = drop_seed;
    hipLaunchKernelGGL((mlp2_fwd_kernel<4>), dim3(cdiv(M, 128)), dim3(256), 0, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_mlp2_bwd_dgrad(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* du, float* dx,
                                    int32_t M, int32_t Cn, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(Cn == 64, MDVIT_E_SHAPE, "mlp2_bwd_dgrad: built for C = 64 (got %d)", Cn);
    MDVIT_CHECK_ARG(M > 0 && Hd > 0 && Hd % 64 == 0, MDVIT_E_SHAPE, "mlp2_bwd_dgrad: need M > 0, hidden %% 64 == 0 (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(gm && x && W1p && b1 && W2tp && W1tp && dx, MDVIT_E_SHAPE, "mlp2_bwd_dgrad: null operand");
    MDVIT_CHECK_ARG(aligned16(gm) && aligned16(x) && aligned16(W1p) && aligned16(b1) && aligned16(W2tp) && aligned16(W1tp) && aligned16(dx) && (!du || aligned16(du)),
                    MDVIT_E_ALIGN, "mlp2_bwd_dgrad: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp2_bwd_dgrad: dropout index space exceeds 2^32");
    Mlp2Args a;
    memset(&a, 0, sizeof(a));
    a.gm = gm; a.x = x; a.W1p = (const uint16_t*)W1p; a.b1 = b1; a.W2tp = (const uint16_t*)W2tp; a.W1tp = (const uint16_t*)W1tp; a.du = du; a.dx = dx;
    a.M = M; a.Hd = Hd;
    a.drop = drop_p > 0.f; a.k1a = key1_0; a.k1b = key1_1;
    a.thresh = (uint32_t)((double)drop_p * 4294967296.0); a.inv_keep = 1.f / (1.f - drop_p);
    a.seed = drop_seed;
    constexpr int smem = 2 * 6 * WTILE;                  // 96 KB
    static int mask = 0;
    const int rc = m2_set_lds(reinterpret_cast<const void*>(&mlp2_bwd_kernel<8>), smem, mask);
    if (rc != MDVIT_OK) return rc;
    hipLaunchKernelGGL((mlp2_bwd_kernel<8>), dim3(cdiv(M, 256)), dim3(512), smem, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
