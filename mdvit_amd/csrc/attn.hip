// Factorized attention core with convolutional relative position encoding and the Domain Adapter
// (FactorAtt_ConvRelPosEnc_Sup.forward, mdvit.py:293-304; ConvRelPosEnc.forward, mpvit.py:296-318),
// forward and backward.  There is no N x N score matrix: softmax runs over the
// TOKEN axis of K per (batch, channel) column, M = softmax(K)^T V is Ch x Ch per head.
//
//   fwd  A: per (token tile, channel group, batch): tile column max / exp-sum / partial K^T V on fp32 MFMA  -> ws
//        B: combine tiles (rescale by exp(m_t - m)) -> M [B,C,Ch], column stats kmax/ksum [B,C]
//        C: U = dwconv_{3|5|7}(v) + bias (LDS-tiled, conv_tile.h; saved);  out = a * (Ch^-0.5 * q.M + q * U)
//   bwd  1: dU = a*G*q (stored), e[b,c] = sum_n G*out as per-workgroup partial rows + fixed-order reduce
//        2: dM = Q^T dFA via the same MFMA tile partials, summed by the batched partial reducer
//        3: crpe weight/bias gradients (tile kernel + finish kernel per window size; mdvit_factoratt_wgrad lets the
//           caller run them on a side stream), conv^T(dU) with the flipped windows
//        4: dq, dk, dv on fp32 MFMA per channel group (t = sum_e dM*M folded into its staging)
// HBM-bound kernels: lanes run along channels (coalesced float4), the Ch x Ch products sit on the matrix cores, token-axis
// reductions are two-stage (partials in a workspace, then a fixed-order combine): deterministic, hundreds of workgroups.
#include "common.h"
#include "conv_tile.h"

namespace {

constexpr int FA_T = 64;      // tokens per tile in the partial (K^T V / Q^T dFA) kernels

// the value of lane ^ 1 (a DPP quad permute [1, 0, 3, 2]: no LDS crossbar, no address register -- __shfl_xor compiles to ds_bpermute_b32)
__device__ __forceinline__ float fa_dpp_xor1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}

struct FaGeom {
    int B, H, W, N, C, heads, Ch, s3, s5, s7;
    float scale;
};

// ---- tile partials: P[c][e] = sum_{n in tile} f(X[n,c]) * Y[n, head(c)*Ch + e] ----------------------
// SOFTMAX: f = exp(x - tile max), also emits tile max / exp-sum (fwd A, X = k, Y = v).
// !SOFTMAX: f = identity, Y scaled per channel by ysc[b,c]*yscale (bwd 2, X = q, Y = G).
// One workgroup = FA_T tokens x one channel group of GW = max(Ch, 32) channels.  Both operand tiles are staged in LDS
// (float4 loads, all in flight before the first LDS store); the GW x GW product X^T Y runs on v_mfma_f32_32x32x2_f32
// with the token axis as k -- each wavefront takes a quarter of the tokens, the four partial results meet in LDS in a
// fixed order -- and only the head-diagonal Ch x Ch blocks are written out.
typedef float fap_f32x16 __attribute__((ext_vector_type(16)));

// Round 4: a workgroup walks NSUB consecutive 64-token tiles and keeps ONE partial result for all of them (online softmax across its tiles: running column
// max / exp-sum in LDS, the accumulators rescaled by exp(m_old - m_new) per tile) -- NSUB times fewer partial rows for the combine pass to walk (it was
// latency-bound on 256 rows per image at the stage-0 shape: 40 us), NSUB times less workspace traffic, the four-wave LDS reduction once per NSUB tiles.
template <int CH, bool SOFTMAX>
__global__ __launch_bounds__(256) void fa_partial_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ Y, long ldy,
                                                         const float* __restrict__ ysc, float yscale,
                                                         float* __restrict__ ws_m, float* __restrict__ ws_s, float* __restrict__ ws_P,
                                                         FaGeom g, int NT, int NSUB,
                                                         const float* __restrict__ outp = nullptr, float* __restrict__ dU = nullptr, float* __restrict__ e_part = nullptr) {
    // (!SOFTMAX, round 4) the backward's first pass rides on this one: X = q and Y = G are exactly what dU = a G q and e = sum_n G out need, so the tile that is
    // staged for dM = Q^T (scale a G) also writes dU (its channel group's columns) and one row of e partials per workgroup -- fa_bwd_prep_kernel's launch and its
    // second read of q and G are gone from the full backward (it stays for the adapter-only call).
    constexpr int GW = CH < 32 ? 32 : CH, NB = (GW + 31) / 32, GQ = GW / 4;
    constexpr int STAGE = 2 * FA_T * GW, RED = 4 * NB * 32 * NB * 32;
    __shared__ __attribute__((aligned(16))) float sm[(STAGE > RED ? STAGE : RED) + 3 * NB * 32 + (SOFTMAX ? 2 * 256 : 0)];
    float* xs = sm;                        // [FA_T][GW]
    float* ys = sm + FA_T * GW;            // [FA_T][GW]
    float* s_m = sm + (STAGE > RED ? STAGE : RED);     // running column max
    float* s_s = s_m + NB * 32;                        // running column exp-sum
    float* s_f = s_s + NB * 32;                        // this tile's rescale factor exp(m_old - m_new)
    float* s_e = s_s;                                  // !SOFTMAX: the workgroup's e sums (the softmax state is not used then)
    float* s_pm = s_f + NB * 32;                       // SOFTMAX: [256 / GW token phases][GW] column maxima of the tile, and
    float* s_ps = s_pm + 256;                          //          the running exp-sums per phase
    const int stile = blockIdx.x, c0 = blockIdx.y * GW, b = blockIdx.z;
    const int NTS = (NT + NSUB - 1) / NSUB;            // partial rows per image
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    fap_f32x16 acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (SOFTMAX && threadIdx.x < NB * 32) { s_m[threadIdx.x] = -INFINITY; s_s[threadIdx.x] = 0.f; s_f[threadIdx.x] = 1.f; }
    if (SOFTMAX) s_ps[threadIdx.x] = 0.f;
    if (!SOFTMAX && e_part && threadIdx.x < NB * 32) s_e[threadIdx.x] = 0.f;
    constexpr int NVE = (FA_T * GQ + 255) / 256;
    float4 eacc[(!SOFTMAX) ? NVE : 1];                  // e sums of this thread's staging slots (slot v always carries channel quad (tid + 256 v) % GQ)
#pragma unroll
    for (int v = 0; v < ((!SOFTMAX) ? NVE : 1); ++v) eacc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    // Round 5: the operands of sub-tile s + 1 are REQUESTED while sub-tile s is reduced (the loop was load -> LDS -> softmax -> MFMA, one exposed HBM round trip per
    // 64 tokens at 2 workgroups per CU: 2.7 TB/s).  Loads are unconditional (clamped row, zeroed afterwards): a load under a branch drains the queue at the join.
    constexpr int NV = (FA_T * GQ + 255) / 256;
    float4 xv[NV], yv[NV], ov[(!SOFTMAX) ? NV : 1];
    auto request = [&](int tile) __attribute__((always_inline)) {
        const int n0r = min(tile, NT - 1) * FA_T;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int i = min((int)threadIdx.x + 256 * v, FA_T * GQ - 1), n = i / GQ, q = i % GQ;
            const long tok = (long)b * g.N + min(n0r + n, g.N - 1);
            xv[v] = *reinterpret_cast<const float4*>(X + tok * ldx + c0 + 4 * q);
            yv[v] = *reinterpret_cast<const float4*>(Y + tok * ldy + c0 + 4 * q);
            if (!SOFTMAX && e_part) ov[v] = *reinterpret_cast<const float4*>(outp + tok * (long)g.C + c0 + 4 * q);
        }
    };
    request(stile * NSUB);
    for (int sub = 0; sub < NSUB; ++sub) {
        const int tile = stile * NSUB + sub;
        if (tile >= NT) break;                         // (uniform)
        const int n0 = tile * FA_T, nt = min(FA_T, g.N - n0);
        // ---- stage (rows past the sequence end are zeros)
        {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = threadIdx.x + 256 * v, n = i / GQ;
                if (!(i < FA_T * GQ && n < nt)) {
                    xv[v] = make_float4(0.f, 0.f, 0.f, 0.f); yv[v] = xv[v];
                    if (!SOFTMAX) ov[v] = xv[v];
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int i = threadIdx.x + 256 * v, n = i / GQ, q = i % GQ;
                if (i < FA_T * GQ) {
                    float4 y4 = yv[v];
                    if (!SOFTMAX) {
                        float4 sc = make_float4(yscale, yscale, yscale, yscale);
                        float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f);
                        if (ysc) { a4 = *reinterpret_cast<const float4*>(ysc + (long)b * g.C + c0 + 4 * q); sc.x *= a4.x; sc.y *= a4.y; sc.z *= a4.z; sc.w *= a4.w; }
                        if (dU && n < nt)          // dU = a G q   (fa_bwd_prep_kernel's product, term for term)
                            *reinterpret_cast<float4*>(dU + ((long)b * g.N + n0 + n) * g.C + c0 + 4 * q) =
                                make_float4(a4.x * y4.x * xv[v].x, a4.y * y4.y * xv[v].y, a4.z * y4.z * xv[v].z, a4.w * y4.w * xv[v].w);
                        if (e_part) {
                            eacc[v].x = fmaf(y4.x, ov[v].x, eacc[v].x); eacc[v].y = fmaf(y4.y, ov[v].y, eacc[v].y);
                            eacc[v].z = fmaf(y4.z, ov[v].z, eacc[v].z); eacc[v].w = fmaf(y4.w, ov[v].w, eacc[v].w);
                        }
                        y4.x *= sc.x; y4.y *= sc.y; y4.z *= sc.z; y4.w *= sc.w;
                    }
                    *reinterpret_cast<float4*>(xs + n * GW + 4 * q) = xv[v];
                    *reinterpret_cast<float4*>(ys + n * GW + 4 * q) = y4;
                }
            }
        }
        if (sub + 1 < NSUB) request(tile + 1);         // in flight across the softmax and the MFMAs of this sub-tile
        __syncthreads();
        if (SOFTMAX) {
            // Column max / exp over the tile's tokens against the RUNNING max.  Round 5: thread (column c = tid % GW, token phase sb = tid / GW) -- a 32-lane group reads 32
            // consecutive columns of ONE token row: no bank conflicts.  (Before: c = tid / 4, sb = tid % 4 -- the four lanes of a column read four rows of the same bank,
            // 4-way conflicts on ~48 LDS operations per thread and tile; the pass was half of the kernel's LDS time, fa_partial<8> at 2.7 TB/s.)  The column maxima of the PER
            // token phases meet in LDS (one more barrier); the exp-sums stay per phase (each rescaled by the same factor) and are added once at the end.
            constexpr int PER = 256 / GW;
            const int c = threadIdx.x % GW, sb = threadIdx.x / GW;
            const bool act = sb < PER;
            float m = -INFINITY;
            if (act)
                for (int n = sb; n < nt; n += PER) m = fmaxf(m, xs[n * GW + c]);
            if (act) s_pm[sb * GW + c] = m;
            __syncthreads();
            float m_old = 0.f, m_new = 0.f;
            if (act) {
                m_old = s_m[c];
                m_new = m_old;
#pragma unroll
                for (int j = 0; j < PER; ++j) m_new = fmaxf(m_new, s_pm[j * GW + c]);
                float ssum = 0.f;
                for (int n = sb; n < nt; n += PER) { const float e = expf(xs[n * GW + c] - m_new); xs[n * GW + c] = e; ssum += e; }
                for (int n = nt + sb; n < FA_T; n += PER) xs[n * GW + c] = 0.f;              // padded rows must not contribute exp(0 - m)
                const float f = expf(m_old - m_new);                                          // 0 on the first tile (m_old = -inf)
                s_ps[sb * GW + c] = fmaf(s_ps[sb * GW + c], f, ssum);
                if (sb == 0) s_f[c] = f;
            }
            __syncthreads();                   // every phase has read s_m / s_pm
            if (act && sb == 0) s_m[c] = m_new;
            if (sub > 0) {                 // rescale what the earlier tiles accumulated: rows c of the D[c][e] blocks
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 f4 = *reinterpret_cast<const float4*>(s_f + 32 * i + 8 * q + 4 * lhi);
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            acc[i][j][4 * q + 0] *= f4.x; acc[i][j][4 * q + 1] *= f4.y; acc[i][j][4 * q + 2] *= f4.z; acc[i][j][4 * q + 3] *= f4.w;
                        }
                    }
            }
        }
        // ---- D[c][e] += X[n][c] * Y[n][e] over this wavefront's quarter of the tokens
        constexpr int TPW = FA_T / 4;
#pragma unroll
        for (int kk = 0; kk < TPW / 2; ++kk) {
            const int n = wave * TPW + 2 * kk + lhi;
            float xa[NB], yb[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) { const int c = 32 * i + l31; xa[i] = c < GW ? xs[n * GW + c] : 0.f; }
#pragma unroll
            for (int j = 0; j < NB; ++j) { const int c = 32 * j + l31; yb[j] = c < GW ? ys[n * GW + c] : 0.f; }
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i], yb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                       // the operand tiles are dead: the next tile may be staged / the partial results may meet in LDS
    }
    if (SOFTMAX && threadIdx.x < GW) {           // (the tile loop ended with a barrier: s_m and the phase sums are final)
        const long o = ((long)b * NTS + stile) * g.C + c0 + threadIdx.x;
        float ssum = 0.f;
#pragma unroll
        for (int j = 0; j < 256 / GW; ++j) ssum += s_ps[j * GW + threadIdx.x];
        ws_m[o] = s_m[threadIdx.x]; ws_s[o] = ssum;
    }
    if (!SOFTMAX && e_part) {                          // (the loop ended with a barrier; s_e was cleared before it)
#pragma unroll
        for (int v = 0; v < NVE; ++v) {
            const int i = threadIdx.x + 256 * v, q = i % GQ;
            if (i < FA_T * GQ) {
                atomicAdd(&s_e[4 * q + 0], eacc[v].x); atomicAdd(&s_e[4 * q + 1], eacc[v].y);
                atomicAdd(&s_e[4 * q + 2], eacc[v].z); atomicAdd(&s_e[4 * q + 3], eacc[v].w);
            }
        }
        __syncthreads();
        if (threadIdx.x < GW) e_part[((long)b * NTS + stile) * g.C + c0 + threadIdx.x] = s_e[threadIdx.x];
        __syncthreads();                               // (s_e aliases nothing the reduction below touches, but keep the phases apart)
    }
    constexpr int RW = NB * 32;            // padded row width of a partial result
    float* red = sm + wave * RW * RW;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 32 * i + 8 * (r >> 2) + 4 * lhi + (r & 3), e = 32 * j + l31;       // lane holds column e, rows c
                red[c * RW + e] = acc[i][j][r];
            }
    __syncthreads();
    for (int o = threadIdx.x; o < GW * CH; o += 256) {
        const int c = o / CH, el = o % CH, e = (c / CH) * CH + el;       // head-diagonal entry (c, e) of the group
        const float v = (sm[c * RW + e] + sm[RW * RW + c * RW + e]) + (sm[2 * RW * RW + c * RW + e] + sm[3 * RW * RW + c * RW + e]);
        ws_P[(((long)b * NTS + stile) * g.C + c0 + c) * CH + el] = v;
    }
}

// ---- the same tile partials at Ch = 8 / 16 (C = 64 / 128) as a STREAM (round 6; see fa_bwd_apply_s8_kernel for the layout) ------------------------------
// C / 4 lanes x float4 cover one token's channels (16 lanes at C = 64: four tokens per wave; 32 at C = 128: two); a lane owns four channels c and keeps P[c][.] (its
// head's Ch columns, local order [mine | lane ^ 1's | lane ^ 2's | lane ^ 3's]) in 4 Ch registers for its whole token run; the other lanes' parts of the head's Y vector
// come by DPP quad permutes (a head is 2 or 4 neighbouring lanes: always inside a quad).  No LDS staging, no MFMA (Ch multiply-adds per element of X), whole rows per load
// instruction.  A workgroup walks the SAME `NSUB` 64-token tiles as fa_partial_kernel and writes the same partial row (the combine kernel / the apply kernel's fixed-order
// sum over the NTS rows do not change); its token slots meet in LDS in slot order.
// SOFTMAX: the running column max is advanced once per group of tokens (one rescale of the accumulators per group); slots merge with exp(m_slot - m).
template <int P> __device__ __forceinline__ float fa_dpp_quad_xor(float x) {
    constexpr int ctrl = P == 1 ? 0xB1 : (P == 2 ? 0x4E : 0x1B);          // quad_perm [1,0,3,2] | [2,3,0,1] | [3,2,1,0]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, true));
}
// (two waves per SIMD by declaration: with the slot rows in DYNAMIC LDS hipcc otherwise plans for four -- 128 registers -- and serialises the token groups' loads behind
//  their arithmetic: the Ch = 8 forward 74 -> 85 us)
template <int CH, bool SOFTMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fa_partial_s8_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ Y, long ldy,
                                                            const float* __restrict__ ysc, float yscale,
                                                            float* __restrict__ ws_m, float* __restrict__ ws_s, float* __restrict__ ws_P,
                                                            FaGeom g, int NT, int NSUB,
                                                            const float* __restrict__ outp, float* __restrict__ dU, float* __restrict__ e_part) {
    constexpr int C = 8 * CH, LPT = C / 4, SLOTS = 256 / LPT, LPH = CH / 4;          // lanes per token row; token slots per workgroup; lanes per head
    constexpr int NACC = 4 * CH + 8;          // per lane: P[4][CH] | (SOFTMAX: m[4], s[4]; else e[4], unused[4])
    // tokens per lane and group (their loads are in flight together; two groups alternate): 4 for the two-operand forward at Ch = 8 (2: 80.9 against 74.1 us at 32 images);
    // 2 for the backward's three operands (160 VGPRs; at 4 it needs 256 -- one wave per SIMD -- for the same 117-123 us) and at Ch = 16 (64 accumulators)
    constexpr int GRP = (SOFTMAX && CH == 8) ? 4 : 2;
    extern __shared__ float fa_s_red[];          // [SLOTS][LPT][NACC + 1]
    const int stile = blockIdx.x, b = blockIdx.z;
    const int NTS = (NT + NSUB - 1) / NSUB;
    const int q = threadIdx.x % LPT, slot = threadIdx.x / LPT;
    const int c0 = 4 * q, hl = q % LPH;          // my four channels; my lane's place inside its head
    const int n_beg = stile * NSUB * FA_T, n_end = min(g.N, n_beg + NSUB * FA_T);
    float acc[4][CH];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int l = 0; l < CH; ++l) acc[j][l] = 0.f;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, ssum[4] = {0.f, 0.f, 0.f, 0.f}, eacc[4] = {0.f, 0.f, 0.f, 0.f};
    float sc[4] = {yscale, yscale, yscale, yscale}, av[4] = {1.f, 1.f, 1.f, 1.f};
    if (!SOFTMAX && ysc) {
        const float4 a4 = *reinterpret_cast<const float4*>(ysc + (long)b * C + c0);
        av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) sc[j] *= av[j];
    }
    struct Rows { float4 x[GRP], y[GRP], o[(!SOFTMAX) ? GRP : 1]; };
    auto request = [&](Rows& r, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const long tok = (long)b * g.N + min(n0 + SLOTS * u, g.N - 1);
            r.x[u] = *reinterpret_cast<const float4*>(X + tok * ldx + c0);
            r.y[u] = *reinterpret_cast<const float4*>(Y + tok * ldy + c0);
            if (!SOFTMAX && e_part) r.o[u] = *reinterpret_cast<const float4*>(outp + tok * (long)C + c0);
        }
    };
    auto compute = [&](const Rows& r, int n0) __attribute__((always_inline)) {
        float xv[GRP][4], yv[GRP][4];
        bool ok[GRP];
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            ok[u] = n0 + SLOTS * u < n_end;
            xv[u][0] = r.x[u].x; xv[u][1] = r.x[u].y; xv[u][2] = r.x[u].z; xv[u][3] = r.x[u].w;
            yv[u][0] = r.y[u].x; yv[u][1] = r.y[u].y; yv[u][2] = r.y[u].z; yv[u][3] = r.y[u].w;
        }
        if (SOFTMAX) {
            float mn[4], f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mn[j] = m[j];
#pragma unroll
                for (int u = 0; u < GRP; ++u) mn[j] = fmaxf(mn[j], ok[u] ? xv[u][j] : -INFINITY);
                f[j] = (mn[j] == -INFINITY) ? 1.f : expf(m[j] - mn[j]);          // (a slot that has seen no token yet keeps its zeros)
                m[j] = mn[j];
                ssum[j] *= f[j];
#pragma unroll
                for (int l = 0; l < CH; ++l) acc[j][l] *= f[j];
            }
#pragma unroll
            for (int u = 0; u < GRP; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[u][j] = ok[u] ? expf(xv[u][j] - mn[j]) : 0.f; ssum[j] += xv[u][j]; }
        }
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            float yh[CH];
            if (!SOFTMAX) {
                if (dU && ok[u]) {          // dU = a G q   (fa_bwd_prep_kernel's product, term for term)
                    const long tok = (long)b * g.N + n0 + SLOTS * u;
                    *reinterpret_cast<float4*>(dU + tok * C + c0) = make_float4(av[0] * yv[u][0] * xv[u][0], av[1] * yv[u][1] * xv[u][1], av[2] * yv[u][2] * xv[u][2], av[3] * yv[u][3] * xv[u][3]);
                }
                if (e_part && ok[u]) {
                    eacc[0] = fmaf(yv[u][0], r.o[u].x, eacc[0]); eacc[1] = fmaf(yv[u][1], r.o[u].y, eacc[1]);
                    eacc[2] = fmaf(yv[u][2], r.o[u].z, eacc[2]); eacc[3] = fmaf(yv[u][3], r.o[u].w, eacc[3]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { yh[j] = ok[u] ? yv[u][j] * sc[j] : 0.f; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) yh[j] = yv[u][j];          // (its weight exp(k - m) is zero past the end)
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                yh[4 + j] = fa_dpp_quad_xor<1>(yh[j]);
                if (CH == 16) { yh[8 + j] = fa_dpp_quad_xor<2>(yh[j]); yh[12 + j] = fa_dpp_quad_xor<3>(yh[j]); }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int l = 0; l < CH; ++l) acc[j][l] = fmaf(xv[u][j], yh[l], acc[j][l]);
        }
    };
    Rows ra, rb;
    int n = n_beg + slot;
    request(ra, n);
    for (; n < n_end; n += 2 * SLOTS * GRP) {
        request(rb, n + SLOTS * GRP);
        compute(ra, n);
        request(ra, n + 2 * SLOTS * GRP);
        compute(rb, n + SLOTS * GRP);
    }
    // ---- the token slots meet in LDS, added in slot order by thread (q, channel j of the quad): one partial row per workgroup
    float* mine = fa_s_red + ((long)slot * LPT + q) * (NACC + 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int l = 0; l < CH; ++l) mine[CH * j + l] = acc[j][l];
        mine[4 * CH + j] = SOFTMAX ? m[j] : eacc[j];
        mine[4 * CH + 4 + j] = ssum[j];
    }
    __syncthreads();
    if (threadIdx.x < C) {
        const int qq = threadIdx.x >> 2, j = threadIdx.x & 3, c = 4 * qq + j;
        const int hq = qq % LPH;
        float tot[CH], t1 = 0.f, mm = -INFINITY;
#pragma unroll
        for (int l = 0; l < CH; ++l) tot[l] = 0.f;
        if (SOFTMAX) {
            for (int sl = 0; sl < SLOTS; ++sl) mm = fmaxf(mm, fa_s_red[((long)sl * LPT + qq) * (NACC + 1) + 4 * CH + j]);
        }
        for (int sl = 0; sl < SLOTS; ++sl) {
            const float* r = fa_s_red + ((long)sl * LPT + qq) * (NACC + 1);
            float f = 1.f;
            if (SOFTMAX) { const float ms = r[4 * CH + j]; f = (ms == -INFINITY) ? 0.f : expf(ms - mm); t1 = fmaf(r[4 * CH + 4 + j], f, t1); }
            else t1 += r[4 * CH + j];
#pragma unroll
            for (int l = 0; l < CH; ++l) tot[l] = SOFTMAX ? fmaf(r[CH * j + l], f, tot[l]) : tot[l] + r[CH * j + l];
        }
        const long orow = ((long)b * NTS + stile) * C + c;
        float* P = ws_P + orow * CH;
#pragma unroll
        for (int l = 0; l < CH; ++l) P[4 * (hq ^ (l >> 2)) + (l & 3)] = tot[l];          // local order -> the head's channel order
        if (SOFTMAX) { ws_m[orow] = mm; ws_s[orow] = t1; }
        else if (e_part) e_part[orow] = t1;
    }
}
template <int CH, bool SOFTMAX>
static void fa_partial_stream_launch(dim3 grid, hipStream_t s, const float* X, long ldx, const float* Y, long ldy, const float* ysc, float yscale, float* ws_m, float* ws_s,
                                     float* ws_P, const FaGeom& g, int NT, int NSUB, const float* outp, float* dU, float* e_part) {
    constexpr int C = 8 * CH, LPT = C / 4, SLOTS = 256 / LPT;
    constexpr int smem = SLOTS * LPT * (4 * CH + 8 + 1) * (int)sizeof(float);
    static bool attr[64] = {false};          // per device: the limit is a property of the function ON a device (one process per GPU is the supported mode; a second device must not inherit the flag)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr[dev] && smem > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_partial_s8_kernel<CH, SOFTMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr[dev] = true;
    }
    hipLaunchKernelGGL((fa_partial_s8_kernel<CH, SOFTMAX>), grid, dim3(256), smem, s, X, ldx, Y, ldy, ysc, yscale, ws_m, ws_s, ws_P, g, NT, NSUB, outp, dU, e_part);
}

// Combine the token-tile partials of the softmax(K)^T V product: online-softmax rescale by exp(m_t - m).  LPO lanes walk the
// NT tiles of one output (c,e) -- 32 for the long token axes of the early stages (a single thread per output paid NT
// dependent loads), 4 or 1 for the late stages' handful of tiles (where 32 lanes per output left 30 of them idle) -- lane
// results folded by a fixed shuffle tree.  (The plain sums of the backward use mdvit_reduce_partials_batched.)
template <int LPO>
__global__ __launch_bounds__(256) void fa_combine_softmax_kernel(const float* __restrict__ ws_m, const float* __restrict__ ws_s,
                                                                 const float* __restrict__ ws_P, float* __restrict__ kmax, float* __restrict__ ksum,
                                                                 float* __restrict__ Mout, FaGeom g, int NT) {
    const int b = blockIdx.y, rl = threadIdx.x % LPO;
    const int o = blockIdx.x * (256 / LPO) + threadIdx.x / LPO;
    if (o >= g.C * g.Ch) return;                       // whole LPO-lane groups leave together
    const int c = o / g.Ch, e = o % g.Ch;
    float m = -INFINITY;
    for (int t = rl; t < NT; t += LPO) m = fmaxf(m, ws_m[((long)b * NT + t) * g.C + c]);
#pragma unroll
    for (int off = LPO / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, LPO));
    float ssum = 0.f, acc = 0.f;
    for (int t = rl; t < NT; t += LPO) {
        const long i = ((long)b * NT + t) * g.C + c;
        const float f = expf(ws_m[i] - m);
        ssum = fmaf(ws_s[i], f, ssum);
        acc = fmaf(ws_P[i * g.Ch + e], f, acc);
    }
#pragma unroll
    for (int off = LPO / 2; off > 0; off >>= 1) { ssum += __shfl_down(ssum, off, LPO); acc += __shfl_down(acc, off, LPO); }
    if (rl == 0) {
        Mout[((long)b * g.C + c) * g.Ch + e] = acc / ssum;
        if (e == 0) { kmax[(long)b * g.C + c] = m; ksum[(long)b * g.C + c] = ssum; }
    }
}

// ---- fwd C': out = a * (Ch^-0.5 * q.M + q * U)    (U from the tiled conv; q rows staged in LDS) -----------
constexpr int FA_OUT_U = 4;
template <int CH>
__global__ __launch_bounds__(512) void fa_out_kernel(const float* __restrict__ qkv, const float* __restrict__ U,
                                                     const float* __restrict__ Mmat, const float* __restrict__ a,
                                                     float* __restrict__ out, FaGeom g, int TLN, int tokens_per_block) {
    extern __shared__ float s_q[];         // [TLN][C]
    const int C = g.C, C3 = 3 * C;
    const int c = threadIdx.x % C, tl = threadIdx.x / C, b = blockIdx.y;
    const int head = c / CH, ch = c % CH, hb = head * CH;
    const float ac = a ? a[(long)b * C + c] : 1.f;
    // this thread's column of the head's matrix, M[hb + j][ch], lives in registers for the whole token run (round 4: it was re-read from the L1 for
    // every token -- CH vector-memory instructions per output element, and the texture-address unit, not HBM, held the kernel: 94 -> 3x us at C = 320)
    float mcol[CH];
    {
        const float* Mb = Mmat + ((long)b * C + hb) * CH + ch;
#pragma unroll
        for (int j = 0; j < CH; ++j) mcol[j] = Mb[(long)j * CH];
    }
    const int n_beg = blockIdx.x * tokens_per_block, n_end = min(g.N, n_beg + tokens_per_block);
    // FA_OUT_U tokens per thread and barrier pair: their loads are in flight together (one token per pair left the kernel latency-bound at ~3 TB/s)
    for (int base = n_beg; base < n_end; base += FA_OUT_U * TLN) {
        float qc[FA_OUT_U], uc[FA_OUT_U];
        long tok[FA_OUT_U];
        bool ok[FA_OUT_U];
#pragma unroll
        for (int u = 0; u < FA_OUT_U; ++u) {
            const int n = base + u * TLN + tl;
            ok[u] = n < n_end;
            tok[u] = (long)b * g.N + (ok[u] ? n : n_beg);
            qc[u] = qkv[tok[u] * C3 + c]; uc[u] = U[tok[u] * C + c];          // (unconditional, clamped: no branch around loads)
        }
#pragma unroll
        for (int u = 0; u < FA_OUT_U; ++u) s_q[(u * TLN + tl) * C + c] = qc[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < FA_OUT_U; ++u) {
            float fa = 0.f;
            const float4* qr = reinterpret_cast<const float4*>(&s_q[(u * TLN + tl) * C + hb]);      // (C and CH are multiples of 4: 16-byte aligned; a wave-wide broadcast read)
#pragma unroll
            for (int j = 0; j < CH / 4; ++j) {                                           // same order of fused multiply-adds as before: same bits
                const float4 q4 = qr[j];
                fa = fmaf(q4.x, mcol[4 * j + 0], fa); fa = fmaf(q4.y, mcol[4 * j + 1], fa);
                fa = fmaf(q4.z, mcol[4 * j + 2], fa); fa = fmaf(q4.w, mcol[4 * j + 3], fa);
            }
            if (ok[u]) out[tok[u] * C + c] = ac * (g.scale * fa + qc[u] * uc[u]);
        }
        __syncthreads();
    }
}

// ---- bwd 1: dU = a*G*q ; e[b,c] = sum_n G[n,c]*out[n,c]  (float4 over channels); e as one partial row per workgroup
__global__ __launch_bounds__(256) void fa_bwd_prep_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                          const float* __restrict__ out, const float* __restrict__ a,
                                                          float* __restrict__ dU, float* __restrict__ e, FaGeom g) {
    extern __shared__ float s_e[];         // [C]
    const int b = blockIdx.y, C = g.C, QC = C >> 2;
    for (int i = threadIdx.x; i < C; i += blockDim.x) s_e[i] = 0.f;
    __syncthreads();
    const long T = (long)gridDim.x * blockDim.x;          // multiple of QC (host guarantees)
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(t0 % QC) * 4;
    float4 av = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a) av = *reinterpret_cast<const float4*>(a + (long)b * C + c);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long total = (long)g.N * QC;
    for (long i = t0; i < total; i += T) {
        const long tok = (long)b * g.N + i / QC;
        const float4 G = *reinterpret_cast<const float4*>(dout + tok * C + c);
        const float4 q = *reinterpret_cast<const float4*>(qkv + tok * 3 * C + c);
        if (dU) *reinterpret_cast<float4*>(dU + tok * C + c) = make_float4(av.x * G.x * q.x, av.y * G.y * q.y, av.z * G.z * q.z, av.w * G.w * q.w);
        if (e) {
            const float4 o = *reinterpret_cast<const float4*>(out + tok * C + c);
            acc.x = fmaf(G.x, o.x, acc.x); acc.y = fmaf(G.y, o.y, acc.y); acc.z = fmaf(G.z, o.z, acc.z); acc.w = fmaf(G.w, o.w, acc.w);
        }
    }
    if (e) {
        atomicAdd(&s_e[c], acc.x); atomicAdd(&s_e[c + 1], acc.y); atomicAdd(&s_e[c + 2], acc.z); atomicAdd(&s_e[c + 3], acc.w);
        __syncthreads();
        for (int i = threadIdx.x; i < C; i += blockDim.x) e[((long)b * gridDim.x + blockIdx.x) * C + i] = s_e[i];   // part [b][block][C]
    }
}

// ---- bwd 5: dq, dk, dv ----------------------------------------------------------------------------
// Per image and head three token x Ch x Ch products (dq = dFA.KV^T, dP = v.dKV^T, dv = P.dKV) plus element-wise terms.
// They run on v_mfma_f32_32x32x2_f32 as D[channel][token] = W[channel][k] . X[k][token]:
//   * a wavefront owns 32 tokens of one channel GROUP (GW = max(Ch, 32) channels = 4/2/1/1 heads for Ch = 8/16/40/64);
//     for Ch < 32 the small matrices are staged block-diagonally in LDS so one 32-wide MFMA tile serves several heads;
//   * lane (token = lane % 32, half = lane / 32) holds its token's channels as the quads {32*(q/4) + 8*(q%4) + 4*half},
//     and the k index of MFMA step kk is mapped to exactly those channels -- so the values a lane loads (float4, once)
//     are both its MFMA operands and the element-wise terms of the 4-consecutive-channel quads the MFMA hands back to it;
//   * W elements come from LDS (row stride GW + 1: conflict-free for row- and column-wise walks).
// The kernel is HBM-bound: reads dout, k, v, U, dVc, writes dq|dk|dv, each exactly once.
typedef float fa_f32x16 __attribute__((ext_vector_type(16)));

// MODE (round 5, A/B by mdvit_factoratt_config): 0 = a tile's rows requested right in front of it (round 4's order, U / dVc moved to the top of the tile);
// 1 = the MFMA operand rows one tile ahead, two waves per SIMD; 2 = the same at ONE wave per SIMD (512 registers: nothing spills)
template <int CH, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CH <= 16 && MODE != 2) ? 2 : 1, (CH <= 16 && MODE != 2) ? 2 : 1))) void fa_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                           const float* __restrict__ U, const float* __restrict__ dVc,
                                                           const float* __restrict__ Mmat, const float* __restrict__ a,
                                                           const float* __restrict__ kmax, const float* __restrict__ ksum,
                                                           const float* __restrict__ dMp, int NTS,
                                                           float* __restrict__ dqkv, FaGeom g, int tiles_per_block) {
    // dMp: the NTS partial rows [B][NTS][C][CH] of dM = Q^T dFA straight from fa_partial_kernel -- summed here in a fixed order while the block-diagonal
    // matrices are staged (round 4: the separate reduction launch per block and sweep is gone; each workgroup re-adds GW x CH x NTS floats, a few KB)
    constexpr int GW = CH < 32 ? 32 : CH;          // channels per group
    constexpr int NB = (GW + 31) / 32;             // 32-channel MFMA row blocks
    constexpr int NQ = GW / 8;                     // float4 quads per lane
    constexpr int KS = GW / 2;                     // MFMA k-steps
    constexpr int LD = GW + 1;
    __shared__ float sKV[GW * LD], sD[GW * LD];
    __shared__ __attribute__((aligned(16))) float s_a[GW], s_km[GW], s_ks[GW], s_tc[GW];
    const int C = g.C, C3 = 3 * C;
    const int b = blockIdx.z, g0 = blockIdx.y * GW;
    for (int i = threadIdx.x; i < GW * GW; i += 256) {
        const int r = i / GW, cc = i % GW;
        const bool same = (r / CH) == (cc / CH);
        const long src = ((long)b * C + g0 + r) * CH + (cc % CH);
        sKV[r * LD + cc] = same ? Mmat[src] : 0.f;
        float dm = 0.f;
        if (same) {                       // eight independent loads at a time (a chain of NTS dependent loads cost a workgroup ~30 us of start-up)
            const float* pp = dMp + (((long)b * NTS) * C + g0 + r) * CH + (cc % CH);
            const long rs = (long)C * CH;
            for (int t0 = 0; t0 < NTS; t0 += 8) {
                float v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int t = t0 + u; v8[u] = pp[(long)(t < NTS ? t : NTS - 1) * rs]; if (t >= NTS) v8[u] = 0.f; }
#pragma unroll
                for (int u = 0; u < 8; ++u) dm += v8[u];
            }
        }
        sD[r * LD + cc] = dm;
    }
    for (int i = threadIdx.x; i < GW; i += 256) {
        const long ci = (long)b * C + g0 + i;
        s_a[i] = a ? a[ci] : 1.f; s_km[i] = kmax[ci]; s_ks[i] = 1.0f / ksum[ci];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < GW; i += 256) {
        float tc = 0.f;                   // t[c] = sum_e dM[c][e] * M[c][e]  (= sum_n P[n,c] dP[n,c], the column-softmax correction)
        const int h0 = (i / CH) * CH;
        for (int e = 0; e < CH; ++e) tc = fmaf(sD[i * LD + h0 + e], sKV[i * LD + h0 + e], tc);
        s_tc[i] = tc;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 31, half = lane >> 5;
    const int ntiles = (g.N + 31) / 32;
    const int tile_beg = blockIdx.x * tiles_per_block, tile_end = min(ntiles, tile_beg + tiles_per_block);
    const float inv_scale = 1.0f / g.scale;
    // Round 5: software prefetch.  The MFMA operand rows of a tile (dout, k, v) are requested one tile AHEAD and its epilogue rows (U, conv^T(dU)) at the top of the tile,
    // all unconditionally (clamped token): they fly under the ~3000 cycles of MFMAs in front of their use; before, every tile paid its own load round trip in front of its
    // first MFMA and a second one (U, dVc -- loaded under `if (ok)`) behind its last: 2.4-2.9 TB/s.  Two register sets alternate (the loop body holds two tiles: no copies).
    struct Rows { float4 g[NQ], k[NQ], v[NQ]; };
    auto request = [&](Rows& r, int tile) __attribute__((always_inline)) {
        const long tok = (long)b * g.N + min(tile * 32 + t, g.N - 1);
        const float* grow = dout + tok * C + g0;
        const float* krow = qkv + tok * C3 + C + g0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int cq = 32 * (q / 4) + 8 * (q % 4) + 4 * half;
            r.g[q] = *reinterpret_cast<const float4*>(grow + cq);
            r.k[q] = *reinterpret_cast<const float4*>(krow + cq);
            r.v[q] = *reinterpret_cast<const float4*>(krow + C + cq);
        }
    };
    auto compute = [&](const Rows& r, int tile) __attribute__((always_inline)) {
        const int n = tile * 32 + t;
        const bool ok = n < g.N;
        const long tok = (long)b * g.N + (ok ? n : g.N - 1);
        float4 ru[NQ], rc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int cq = 32 * (q / 4) + 8 * (q % 4) + 4 * half;
            ru[q] = *reinterpret_cast<const float4*>(U + tok * C + g0 + cq);
            rc[q] = *reinterpret_cast<const float4*>(dVc + tok * C + g0 + cq);
        }
        float dfa[NQ][4], vv[NQ][4], pp[NQ][4];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int cq = 32 * (q / 4) + 8 * (q % 4) + 4 * half;
            const float4 g4 = r.g[q], k4 = r.k[q], v4 = r.v[q];
            const float4 a4 = *reinterpret_cast<const float4*>(s_a + cq);
            const float4 m4 = *reinterpret_cast<const float4*>(s_km + cq);
            const float4 i4 = *reinterpret_cast<const float4*>(s_ks + cq);
            const float z = ok ? 1.f : 0.f;          // out-of-range tokens contribute zeros (their results are not stored)
            dfa[q][0] = z * g.scale * a4.x * g4.x; dfa[q][1] = z * g.scale * a4.y * g4.y;
            dfa[q][2] = z * g.scale * a4.z * g4.z; dfa[q][3] = z * g.scale * a4.w * g4.w;
            vv[q][0] = z * v4.x; vv[q][1] = z * v4.y; vv[q][2] = z * v4.z; vv[q][3] = z * v4.w;
            pp[q][0] = z * expf(k4.x - m4.x) * i4.x; pp[q][1] = z * expf(k4.y - m4.y) * i4.y;
            pp[q][2] = z * expf(k4.z - m4.z) * i4.z; pp[q][3] = z * expf(k4.w - m4.w) * i4.w;
        }
        fa_f32x16 acc1[NB], acc2[NB], acc3[NB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) { acc1[ob][rr] = 0.f; acc2[ob][rr] = 0.f; acc3[ob][rr] = 0.f; }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const int orow = 32 * ob + t;                       // W row (output channel) this lane feeds
            const bool rok = orow < GW;
            const int orc = rok ? orow : 0;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int q = kk / 4, j = kk % 4;
                const int kc = 32 * (q / 4) + 8 * (q % 4) + 4 * half + j;      // channel of this lane's k slot
                float w1 = sKV[orc * LD + kc], w2 = sD[orc * LD + kc], w3 = sD[kc * LD + orc];
                if (!rok) { w1 = 0.f; w2 = 0.f; w3 = 0.f; }
                acc1[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, dfa[q][j], acc1[ob], 0, 0, 0);
                acc2[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2, vv[q][j], acc2[ob], 0, 0, 0);
                acc3[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w3, pp[q][j], acc3[ob], 0, 0, 0);
            }
        }
        if (ok) {
            float* drow = dqkv + tok * C3 + g0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ob = q / 4, r0 = 4 * (q % 4);
                const int cq = 32 * (q / 4) + 8 * (q % 4) + 4 * half;
                const float4 u4 = ru[q], c4 = rc[q];
                const float4 t4 = *reinterpret_cast<const float4*>(s_tc + cq);
                float4 dq, dk, dv;
                dq.x = fmaf(dfa[q][0] * inv_scale, u4.x, acc1[ob][r0 + 0]); dq.y = fmaf(dfa[q][1] * inv_scale, u4.y, acc1[ob][r0 + 1]);
                dq.z = fmaf(dfa[q][2] * inv_scale, u4.z, acc1[ob][r0 + 2]); dq.w = fmaf(dfa[q][3] * inv_scale, u4.w, acc1[ob][r0 + 3]);
                dk.x = pp[q][0] * (acc2[ob][r0 + 0] - t4.x); dk.y = pp[q][1] * (acc2[ob][r0 + 1] - t4.y);
                dk.z = pp[q][2] * (acc2[ob][r0 + 2] - t4.z); dk.w = pp[q][3] * (acc2[ob][r0 + 3] - t4.w);
                dv.x = acc3[ob][r0 + 0] + c4.x; dv.y = acc3[ob][r0 + 1] + c4.y;
                dv.z = acc3[ob][r0 + 2] + c4.z; dv.w = acc3[ob][r0 + 3] + c4.w;
                *reinterpret_cast<float4*>(drow + cq) = dq;
                *reinterpret_cast<float4*>(drow + C + cq) = dk;
                *reinterpret_cast<float4*>(drow + 2 * C + cq) = dv;
            }
        }
    };
    Rows ra, rb;
    int tile = tile_beg + wave;
    if (MODE == 0) {
        for (; tile < tile_end; tile += 4) { request(ra, tile); compute(ra, tile); }
        return;
    }
    if (tile < tile_end) request(ra, tile);
    for (; tile < tile_end; tile += 8) {
        request(rb, min(tile + 4, ntiles - 1));
        compute(ra, tile);
        if (tile + 4 < tile_end) {
            request(ra, min(tile + 8, ntiles - 1));
            compute(rb, tile + 4);
        }
    }
}

// ---- bwd 5 at Ch = 8, C = 64 as a STREAM (round 6) --------------------------------------------------------------------------------------------
// fa_bwd_apply_kernel<8> runs the three token x 8 x 8 products of every head on 32 x 32 fp32 MFMA tiles of block-diagonal matrices (15/16 of every tile is zero) with
// operand quads of 32 bytes per token and load instruction -- 32 cache lines per wave-wide load -- and sits at 3.3-3.9 TB/s of its eight [tokens, C] streams.  At Ch = 8
// the products are 24 fused multiply-adds per output element: cheap enough for the VALU (the arithmetic below prices out at ~6x the HBM rate), so here the layout is
// chosen for the MEMORY system instead: 16 lanes x float4 cover one token's 64 channels (every load / store instruction moves whole 256-byte rows, four tokens per wave),
// a lane keeps the matrix rows / columns of ITS four channels in registers for its whole token run (96 values: M[c][.], dM[c][.], dM[.][c]) and gets the other half of
// its head's 8-vector from the neighbouring lane with three 4-value DPP exchanges per token.  The next token's five rows are requested before the current one is used.
// Same products; the sums run in the head's local channel order instead of the MFMA's (fp32 round-off only: tests compare against fp64 at 1e-4).
__global__ __launch_bounds__(256) void fa_bwd_apply_s8_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                              const float* __restrict__ U, const float* __restrict__ dVc,
                                                              const float* __restrict__ Mmat, const float* __restrict__ a,
                                                              const float* __restrict__ kmax, const float* __restrict__ ksum,
                                                              const float* __restrict__ dMp, int NTS,
                                                              float* __restrict__ dqkv, FaGeom g, int tokens_per_block) {
    constexpr int C = 64, CH = 8, C3 = 192;
    __shared__ float sM[C * CH], sD[C * CH];
    __shared__ float s_tc[C];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C * CH; i += 256) {
        sM[i] = Mmat[(long)b * C * CH + i];
        float dm = 0.f;
        const float* pp = dMp + ((long)b * NTS) * C * CH + i;
        const long rs = (long)C * CH;
        for (int t0 = 0; t0 < NTS; t0 += 8) {          // (the summation order of fa_bwd_apply_kernel: eight independent loads at a time)
            float v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int t = t0 + u; v8[u] = pp[(long)(t < NTS ? t : NTS - 1) * rs]; if (t >= NTS) v8[u] = 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) dm += v8[u];
        }
        sD[i] = dm;
    }
    __syncthreads();
    if (threadIdx.x < C) {
        float tc = 0.f;
#pragma unroll
        for (int e = 0; e < CH; ++e) tc = fmaf(sD[threadIdx.x * CH + e], sM[threadIdx.x * CH + e], tc);
        s_tc[threadIdx.x] = tc;
    }
    __syncthreads();
    const int q = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int c0 = 4 * q, hb = (q >> 1) * CH, j0 = 4 * (q & 1), jo = j0 ^ 4;          // my four channels; my head; my / my neighbour's offset inside the head
    // local order of a head's 8-vector: [mine (4) | the neighbour lane's (4)]
    float Mr[4][8], Dr[4][8], Dc[8][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const int e = l < 4 ? j0 + l : jo + (l - 4);
            Mr[j][l] = sM[(c0 + j) * CH + e];          // dq[c] = sum_e dfa[e] M[c][e]
            Dr[j][l] = sD[(c0 + j) * CH + e];          // dk[c] = P[c] (sum_e v[e] dM[c][e] - t[c])
            Dc[l][j] = sD[(hb + e) * CH + j0 + j];     // dv[e'] = sum_c P[c] dM[c][e'],  e' = my channel j, c = the head's channel e
        }
    const long ci = (long)b * C + c0;
    const float4 a4 = a ? *reinterpret_cast<const float4*>(a + ci) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 km4 = *reinterpret_cast<const float4*>(kmax + ci);
    float4 is4 = *reinterpret_cast<const float4*>(ksum + ci);
    is4 = make_float4(1.0f / is4.x, 1.0f / is4.y, 1.0f / is4.z, 1.0f / is4.w);
    const float4 tc4 = *reinterpret_cast<const float4*>(s_tc + c0);
    const float inv_scale = 1.0f / g.scale;
    const int n_beg = blockIdx.x * tokens_per_block, n_end = min(g.N, n_beg + tokens_per_block);
    struct Rows { float4 g, k, v, u, c; };
    auto request = [&](Rows& r, int n) __attribute__((always_inline)) {
        const long tok = (long)b * g.N + min(n, g.N - 1);
        r.g = *reinterpret_cast<const float4*>(dout + tok * C + c0);
        r.k = *reinterpret_cast<const float4*>(qkv + tok * C3 + C + c0);
        r.v = *reinterpret_cast<const float4*>(qkv + tok * C3 + 2 * C + c0);
        r.u = *reinterpret_cast<const float4*>(U + tok * C + c0);
        r.c = *reinterpret_cast<const float4*>(dVc + tok * C + c0);
    };
    auto compute = [&](const Rows& r, int n) __attribute__((always_inline)) {
        float dfa[8], vv[8], pp[8];
        dfa[0] = g.scale * a4.x * r.g.x; dfa[1] = g.scale * a4.y * r.g.y; dfa[2] = g.scale * a4.z * r.g.z; dfa[3] = g.scale * a4.w * r.g.w;
        vv[0] = r.v.x; vv[1] = r.v.y; vv[2] = r.v.z; vv[3] = r.v.w;
        pp[0] = expf(r.k.x - km4.x) * is4.x; pp[1] = expf(r.k.y - km4.y) * is4.y; pp[2] = expf(r.k.z - km4.z) * is4.z; pp[3] = expf(r.k.w - km4.w) * is4.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dfa[4 + j] = fa_dpp_xor1(dfa[j]); vv[4 + j] = fa_dpp_xor1(vv[j]); pp[4 + j] = fa_dpp_xor1(pp[j]); }
        float dq[4], dk[4], dv[4];
        const float uu[4] = {r.u.x, r.u.y, r.u.z, r.u.w}, cc[4] = {r.c.x, r.c.y, r.c.z, r.c.w}, tcv[4] = {tc4.x, tc4.y, tc4.z, tc4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l) { s1 = fmaf(dfa[l], Mr[j][l], s1); s2 = fmaf(vv[l], Dr[j][l], s2); s3 = fmaf(pp[l], Dc[l][j], s3); }
            dq[j] = fmaf(dfa[j] * inv_scale, uu[j], s1);
            dk[j] = pp[j] * (s2 - tcv[j]);
            dv[j] = s3 + cc[j];
        }
        if (n < n_end) {
            float* drow = dqkv + ((long)b * g.N + n) * C3 + c0;
            *reinterpret_cast<float4*>(drow) = make_float4(dq[0], dq[1], dq[2], dq[3]);
            *reinterpret_cast<float4*>(drow + C) = make_float4(dk[0], dk[1], dk[2], dk[3]);
            *reinterpret_cast<float4*>(drow + 2 * C) = make_float4(dv[0], dv[1], dv[2], dv[3]);
        }
    };
    // (every lane of a 16-lane row group runs the same trip count: the DPP exchange partners are always alive)
    Rows ra, rb;
    int n = n_beg + slot;
    request(ra, n);
    for (; n < n_end; n += 32) {
        request(rb, n + 16);
        compute(ra, n);
        request(ra, n + 32);
        compute(rb, n + 16);
    }
}

// ---- bwd 5 at Ch = 16, C = 128 as a stream (round 6) ---------------------------------------------------------------------------------------------
// The layout of fa_bwd_apply_s8_kernel with 32 lanes x float4 per token (two tokens per wave) and a head on the four lanes of a quad.  A lane's matrix rows / columns are
// 192 values here -- more than the register file leaves next to two token sets in flight -- so they live in an LDS table [48 float4 slots][32 lane classes]: a wave-wide
// ds_read_b128 of one slot is 512 contiguous bytes (conflict-free; the second token's lanes read the same addresses: broadcast).  48 reads per token and lane price out at
// ~2.6x the HBM rate of the kernel's eight streams; the three 16-vectors of the head come by quad-permute DPP.
template <int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fa_bwd_apply_tab_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                               const float* __restrict__ U, const float* __restrict__ dVc,
                                                               const float* __restrict__ Mmat, const float* __restrict__ a,
                                                               const float* __restrict__ kmax, const float* __restrict__ ksum,
                                                               const float* __restrict__ dMp, int NTS,
                                                               float* __restrict__ dqkv, FaGeom g, int tokens_per_block) {
    constexpr int C = 8 * CH, C3 = 3 * C, LPT = C / 4, SLOTS = 256 / LPT, LPH = CH / 4, L4 = CH / 4, NS = 8 * L4 + CH;          // table slots: M rows, dM rows (4 j x L4), dM columns (CH)
    __shared__ __attribute__((aligned(16))) float sM[C * CH], sD[C * CH];
    __shared__ __attribute__((aligned(16))) float4 tab[NS][LPT];
    __shared__ float s_tc[C];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C * CH; i += 256) {
        sM[i] = Mmat[(long)b * C * CH + i];
        float dm = 0.f;
        const float* pp = dMp + ((long)b * NTS) * C * CH + i;
        const long rs = (long)C * CH;
        for (int t0 = 0; t0 < NTS; t0 += 8) {          // (the summation order of fa_bwd_apply_kernel: eight independent loads at a time)
            float v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int t = t0 + u; v8[u] = pp[(long)(t < NTS ? t : NTS - 1) * rs]; if (t >= NTS) v8[u] = 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) dm += v8[u];
        }
        sD[i] = dm;
    }
    __syncthreads();
    if (threadIdx.x < C) {
        float tc = 0.f;
#pragma unroll
        for (int e = 0; e < CH; ++e) tc = fmaf(sD[threadIdx.x * CH + e], sM[threadIdx.x * CH + e], tc);
        s_tc[threadIdx.x] = tc;
    }
    // the table: slots [0, 4 L4) M[c0 + j][local 4 l4 ..], [4 L4, 8 L4) dM[c0 + j][local 4 l4 ..] (slot = L4 j + l4), then CH slots dM[head's channel e(l)][my four columns];
    // local index l <-> the head's channel e(l) = 4 ((q % LPH) ^ (l / 4)) + l % 4
    for (int i = threadIdx.x; i < NS * LPT; i += 256) {
        const int slot = i / LPT, qq = i % LPT, hl = qq % LPH, hb = (qq / LPH) * CH, cc0 = 4 * qq;
        float4 v;
        if (slot < 8 * L4) {
            const int sl = slot % (4 * L4);
            const float* src = (slot < 4 * L4 ? sM : sD) + (cc0 + sl / L4) * CH + 4 * (hl ^ (sl % L4));
            v = *reinterpret_cast<const float4*>(src);
        } else {
            const int l = slot - 8 * L4, e = 4 * (hl ^ (l >> 2)) + (l & 3);
            v = *reinterpret_cast<const float4*>(sD + (hb + e) * CH + 4 * hl);
        }
        tab[slot][qq] = v;
    }
    __syncthreads();
    const int q = threadIdx.x % LPT, slot = threadIdx.x / LPT;
    const int c0 = 4 * q;
    const long ci = (long)b * C + c0;
    const float4 a4 = a ? *reinterpret_cast<const float4*>(a + ci) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 km4 = *reinterpret_cast<const float4*>(kmax + ci);
    float4 is4 = *reinterpret_cast<const float4*>(ksum + ci);
    is4 = make_float4(1.0f / is4.x, 1.0f / is4.y, 1.0f / is4.z, 1.0f / is4.w);
    const float4 tc4 = *reinterpret_cast<const float4*>(s_tc + c0);
    const float inv_scale = 1.0f / g.scale;
    const int n_beg = blockIdx.x * tokens_per_block, n_end = min(g.N, n_beg + tokens_per_block);
    struct Rows { float4 g, k, v, u, c; };
    auto request = [&](Rows& r, int n) __attribute__((always_inline)) {
        const long tok = (long)b * g.N + min(n, g.N - 1);
        r.g = *reinterpret_cast<const float4*>(dout + tok * C + c0);
        r.k = *reinterpret_cast<const float4*>(qkv + tok * C3 + C + c0);
        r.v = *reinterpret_cast<const float4*>(qkv + tok * C3 + 2 * C + c0);
        r.u = *reinterpret_cast<const float4*>(U + tok * C + c0);
        r.c = *reinterpret_cast<const float4*>(dVc + tok * C + c0);
    };
    auto compute = [&](const Rows& r, int n) __attribute__((always_inline)) {
        float dfa[CH], vv[CH], pp[CH];
        dfa[0] = g.scale * a4.x * r.g.x; dfa[1] = g.scale * a4.y * r.g.y; dfa[2] = g.scale * a4.z * r.g.z; dfa[3] = g.scale * a4.w * r.g.w;
        vv[0] = r.v.x; vv[1] = r.v.y; vv[2] = r.v.z; vv[3] = r.v.w;
        pp[0] = expf(r.k.x - km4.x) * is4.x; pp[1] = expf(r.k.y - km4.y) * is4.y; pp[2] = expf(r.k.z - km4.z) * is4.z; pp[3] = expf(r.k.w - km4.w) * is4.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dfa[4 + j] = fa_dpp_quad_xor<1>(dfa[j]); vv[4 + j] = fa_dpp_quad_xor<1>(vv[j]); pp[4 + j] = fa_dpp_quad_xor<1>(pp[j]);
            if constexpr (CH == 16) {
                dfa[8 + j] = fa_dpp_quad_xor<2>(dfa[j]); dfa[12 + j] = fa_dpp_quad_xor<3>(dfa[j]);
                vv[8 + j] = fa_dpp_quad_xor<2>(vv[j]); vv[12 + j] = fa_dpp_quad_xor<3>(vv[j]);
                pp[8 + j] = fa_dpp_quad_xor<2>(pp[j]); pp[12 + j] = fa_dpp_quad_xor<3>(pp[j]);
            }
        }
        float dq[4], dk[4], dv[4] = {0.f, 0.f, 0.f, 0.f};
        const float uu[4] = {r.u.x, r.u.y, r.u.z, r.u.w}, cc[4] = {r.c.x, r.c.y, r.c.z, r.c.w}, tcv[4] = {tc4.x, tc4.y, tc4.z, tc4.w};
        // (the table is re-read from LDS for every token: hoisted out of the token loop -- 192 registers -- it spills)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int l4 = 0; l4 < L4; ++l4) {
                const float4 m4 = tab[L4 * j + l4][q], d4 = tab[4 * L4 + L4 * j + l4][q];
                s1 = fmaf(dfa[4 * l4 + 0], m4.x, s1); s1 = fmaf(dfa[4 * l4 + 1], m4.y, s1); s1 = fmaf(dfa[4 * l4 + 2], m4.z, s1); s1 = fmaf(dfa[4 * l4 + 3], m4.w, s1);
                s2 = fmaf(vv[4 * l4 + 0], d4.x, s2); s2 = fmaf(vv[4 * l4 + 1], d4.y, s2); s2 = fmaf(vv[4 * l4 + 2], d4.z, s2); s2 = fmaf(vv[4 * l4 + 3], d4.w, s2);
            }
            dq[j] = fmaf(dfa[j] * inv_scale, uu[j], s1);
            dk[j] = pp[j] * (s2 - tcv[j]);
        }
#pragma unroll
        for (int l = 0; l < CH; ++l) {
            const float4 dc = tab[8 * L4 + l][q];
            dv[0] = fmaf(pp[l], dc.x, dv[0]); dv[1] = fmaf(pp[l], dc.y, dv[1]); dv[2] = fmaf(pp[l], dc.z, dv[2]); dv[3] = fmaf(pp[l], dc.w, dv[3]);
        }
        if (n < n_end) {
            float* drow = dqkv + ((long)b * g.N + n) * C3 + c0;
            *reinterpret_cast<float4*>(drow) = make_float4(dq[0], dq[1], dq[2], dq[3]);
            *reinterpret_cast<float4*>(drow + C) = make_float4(dk[0], dk[1], dk[2], dk[3]);
            *reinterpret_cast<float4*>(drow + 2 * C) = make_float4(dv[0] + cc[0], dv[1] + cc[1], dv[2] + cc[2], dv[3] + cc[3]);
        }
    };
    Rows ra, rb;
    int n = n_beg + slot;
    request(ra, n);
    for (; n < n_end; n += 2 * SLOTS) {
        request(rb, n + SLOTS);
        compute(ra, n);
        request(ra, n + 2 * SLOTS);
        compute(rb, n + SLOTS);
    }
}

// role 0: dq = a*scale*G . KV^T (+ G*a*U);  1: dk = P * (v . dM^T - t);  2: dv = P . dM + conv^T(dU)      -- fa_bwd_apply3_kernel's wave roles
template <int CH, int ROLE>
__device__ __forceinline__ void fa_apply3_tiles(const float* __restrict__ dout, const float* __restrict__ qkv, const float* __restrict__ U, const float* __restrict__ dVc,
                                                float* __restrict__ dqkv, const FaGeom& g, const float* sKV, const float* sD, const float* s_a, const float* s_km,
                                                const float* s_ks, const float* s_tc, int b, int g0, int slot, int tiles_per_block, int bx) {
    constexpr int GW = CH, NB = (GW + 31) / 32, NQ = GW / 8, KS = GW / 2, LD = GW + 1;
    const int C = g.C, C3 = 3 * C;
    const int lane = threadIdx.x & 63;
    const int t = lane & 31, half = lane >> 5;
    const int ntiles = (g.N + 31) / 32;
    const int tile_beg = bx * tiles_per_block, tile_end = min(ntiles, tile_beg + tiles_per_block);
    const float inv_scale = 1.0f / g.scale;
    // Round 5: the tile's two operand rows are requested one tile AHEAD (unconditional, clamped token) and fly under the MFMAs of the tile in front of them -- the per-tile chain
    // load -> exp -> CH / 2 dependent MFMAs -> store had nothing to overlap with (profiles/r04_fa_bwd_apply3_pmc_stalls.txt: a wave waited 65 % of its life)
    struct Rows { float4 a[NQ], b[NQ]; };
    auto request = [&](Rows& r, int tile) __attribute__((always_inline)) {
        const long tok = (long)b * g.N + min(tile * 32 + t, g.N - 1);
        // A: dout | k | k;  B: U | v | conv^T(dU)
        const float* arow = ROLE == 0 ? dout + tok * C + g0 : qkv + tok * C3 + C + g0;
        const float* brow = ROLE == 0 ? U + tok * C + g0 : (ROLE == 1 ? qkv + tok * C3 + 2 * C + g0 : dVc + tok * C + g0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int cq = 32 * (q / 4) + 8 * (q % 4) + 4 * half;
            r.a[q] = *reinterpret_cast<const float4*>(arow + cq);
            r.b[q] = *reinterpret_cast<const float4*>(brow + cq);
        }
    };
    auto compute = [&](const Rows& r, int tile) __attribute__((always_inline)) {
        const int n = tile * 32 + t;
        const bool ok = n < g.N;
        const float z = ok ? 1.f : 0.f;                  // out-of-range tokens contribute zeros (their results are not stored)
        const long tok = (long)b * g.N + (ok ? n : g.N - 1);
        float xo[NQ][4], el[NQ][4];                      // the product's token operand; its element-wise companion
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int cq = 32 * (q / 4) + 8 * (q % 4) + 4 * half;
            const float4 a4 = r.a[q], b4 = r.b[q];
            if (ROLE == 0) {
                const float4 s4 = *reinterpret_cast<const float4*>(s_a + cq);
                xo[q][0] = z * g.scale * s4.x * a4.x; xo[q][1] = z * g.scale * s4.y * a4.y;
                xo[q][2] = z * g.scale * s4.z * a4.z; xo[q][3] = z * g.scale * s4.w * a4.w;
                el[q][0] = b4.x; el[q][1] = b4.y; el[q][2] = b4.z; el[q][3] = b4.w;
            } else {
                const float4 m4 = *reinterpret_cast<const float4*>(s_km + cq);
                const float4 i4 = *reinterpret_cast<const float4*>(s_ks + cq);
                const float p0 = z * expf(a4.x - m4.x) * i4.x, p1 = z * expf(a4.y - m4.y) * i4.y;
                const float p2 = z * expf(a4.z - m4.z) * i4.z, p3 = z * expf(a4.w - m4.w) * i4.w;
                if (ROLE == 1) {
                    xo[q][0] = z * b4.x; xo[q][1] = z * b4.y; xo[q][2] = z * b4.z; xo[q][3] = z * b4.w;
                    el[q][0] = p0; el[q][1] = p1; el[q][2] = p2; el[q][3] = p3;
                } else {
                    xo[q][0] = p0; xo[q][1] = p1; xo[q][2] = p2; xo[q][3] = p3;
                    el[q][0] = b4.x; el[q][1] = b4.y; el[q][2] = b4.z; el[q][3] = b4.w;
                }
            }
        }
        fa_f32x16 acc[NB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) acc[ob][rr] = 0.f;
        // (the matrix elements are re-read from LDS for every tile: hoisted out of the tile loop -- NB x CH / 2 registers -- they and the prefetched rows do not fit)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const int orow = 32 * ob + t;
            const bool rok = orow < GW;
            const int orc = rok ? orow : 0;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int q = kk / 4, j = kk % 4;
                const int kc = 32 * (q / 4) + 8 * (q % 4) + 4 * half + j;
                float w = ROLE == 0 ? sKV[orc * LD + kc] : (ROLE == 1 ? sD[orc * LD + kc] : sD[kc * LD + orc]);     // W[orow][kc]: KV | dM | dM^T
                if (!rok) w = 0.f;
                acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, xo[q][j], acc[ob], 0, 0, 0);
            }
        }
        if (ok) {
            float* drow = dqkv + tok * C3 + ROLE * C + g0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ob = q / 4, r0 = 4 * (q % 4);
                const int cq = 32 * (q / 4) + 8 * (q % 4) + 4 * half;
                float4 d;
                if (ROLE == 0) {
                    d.x = fmaf(xo[q][0] * inv_scale, el[q][0], acc[ob][r0 + 0]); d.y = fmaf(xo[q][1] * inv_scale, el[q][1], acc[ob][r0 + 1]);
                    d.z = fmaf(xo[q][2] * inv_scale, el[q][2], acc[ob][r0 + 2]); d.w = fmaf(xo[q][3] * inv_scale, el[q][3], acc[ob][r0 + 3]);
                } else if (ROLE == 1) {
                    const float4 t4 = *reinterpret_cast<const float4*>(s_tc + cq);
                    d.x = el[q][0] * (acc[ob][r0 + 0] - t4.x); d.y = el[q][1] * (acc[ob][r0 + 1] - t4.y);
                    d.z = el[q][2] * (acc[ob][r0 + 2] - t4.z); d.w = el[q][3] * (acc[ob][r0 + 3] - t4.w);
                } else {
                    d.x = acc[ob][r0 + 0] + el[q][0]; d.y = acc[ob][r0 + 1] + el[q][1];
                    d.z = acc[ob][r0 + 2] + el[q][2]; d.w = acc[ob][r0 + 3] + el[q][3];
                }
                *reinterpret_cast<float4*>(drow + cq) = d;
            }
        }
    };
    Rows ra, rb;
    int tile = tile_beg + slot;
    if (tile < tile_end) request(ra, tile);
    for (; tile < tile_end; tile += 4) {
        request(rb, min(tile + 2, ntiles - 1));
        compute(ra, tile);
        if (tile + 2 < tile_end) {
            request(ra, min(tile + 4, ntiles - 1));
            compute(rb, tile + 2);
        }
    }
}

// The same three products for Ch >= 32 (one head per channel group) with the PRODUCTS dealt to wavefronts: wave role 0 forms dq, 1 dk, 2 dv of a 32-token
// tile -- each with one accumulator set and the two operands its product needs (~1/3 of the registers: three waves per SIMD instead of the one that the
// all-in-one wave above gets at Ch = 40 / 64, where 396 / 512 registers left the loads of a tile nothing to hide behind: 1.9 TB/s).  k is read by two roles
// (the second read hits the L2).  Per output element the same MFMA sequence as above: same bits.
template <int CH>
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(CH >= 64 ? 2 : 3, CH >= 64 ? 2 : 3))) void fa_bwd_apply3_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                           const float* __restrict__ U, const float* __restrict__ dVc,
                                                           const float* __restrict__ Mmat, const float* __restrict__ a,
                                                           const float* __restrict__ kmax, const float* __restrict__ ksum,
                                                           const float* __restrict__ dMp, int NTS,
                                                           float* __restrict__ dqkv, FaGeom g, int tiles_per_block, int gx, int xcd_map) {
    static_assert(CH >= 32 && CH % 8 == 0, "one head per channel group");
    constexpr int GW = CH, NB = (GW + 31) / 32, NQ = GW / 8, KS = GW / 2, LD = GW + 1, NTH = 384;
    __shared__ float sKV[GW * LD], sD[GW * LD];
    __shared__ __attribute__((aligned(16))) float s_a[GW], s_km[GW], s_ks[GW], s_tc[GW];
    const int C = g.C, C3 = 3 * C;
    // Workgroup -> (token range bx, head by, image bz).  A head's operand rows are Ch-float pieces of [tokens, C] rows: at Ch = 40 a 160-byte piece straddles two 128-byte
    // lines, each shared with the neighbouring head.  Dealt (x, y, z) the heads of one token range land on two XCDs alternately (round-robin dispatch) and every L2 fetches
    // its own copy of the shared lines: the PMC passes count 12.8 [tokens, C] reads per launch where the operands are 6-7 (profiles/pmc_traffic.json).  xcd_map: the launch
    // is one-dimensional and the heads of a token range get CONSECUTIVE slots of ONE XCD (workgroup i runs on XCD i % 8: slot = (i % 8) * (n / 8) + i / 8).
    int bx, by, bz;
    if (xcd_map) {
        const int nwg = gridDim.x, heads = C / GW;
        const int i = blockIdx.x, slot = (i & 7) * (nwg >> 3) + (i >> 3);
        by = slot % heads; bx = (slot / heads) % gx; bz = slot / (heads * gx);
    } else { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; }
    const int b = bz, g0 = by * GW;
    for (int i = threadIdx.x; i < GW * GW; i += NTH) {
        const int r = i / GW, cc = i % GW;
        sKV[r * LD + cc] = Mmat[((long)b * C + g0 + r) * CH + cc];
        float dm = 0.f;
        const float* pp = dMp + (((long)b * NTS) * C + g0 + r) * CH + cc;
        const long rs = (long)C * CH;
        for (int t0 = 0; t0 < NTS; t0 += 8) {          // (the summation order of fa_bwd_apply_kernel)
            float v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int t = t0 + u; v8[u] = pp[(long)(t < NTS ? t : NTS - 1) * rs]; if (t >= NTS) v8[u] = 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) dm += v8[u];
        }
        sD[r * LD + cc] = dm;
    }
    for (int i = threadIdx.x; i < GW; i += NTH) {
        const long ci = (long)b * C + g0 + i;
        s_a[i] = a ? a[ci] : 1.f; s_km[i] = kmax[ci]; s_ks[i] = 1.0f / ksum[ci];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < GW; i += NTH) {
        float tc = 0.f;
        for (int e = 0; e < CH; ++e) tc = fmaf(sD[i * LD + e], sKV[i * LD + e], tc);
        s_tc[i] = tc;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const int role = wave % 3, slot = wave / 3;
    // one straight-line tile loop per role (a role test INSIDE the loop put every load under a branch, and the compiler drains the vector-memory queue at
    // each join: five serialised load round trips per tile)
    if (role == 0) fa_apply3_tiles<CH, 0>(dout, qkv, U, dVc, dqkv, g, sKV, sD, s_a, s_km, s_ks, s_tc, b, g0, slot, tiles_per_block, bx);
    else if (role == 1) fa_apply3_tiles<CH, 1>(dout, qkv, U, dVc, dqkv, g, sKV, sD, s_a, s_km, s_ks, s_tc, b, g0, slot, tiles_per_block, bx);
    else fa_apply3_tiles<CH, 2>(dout, qkv, U, dVc, dqkv, g, sKV, sD, s_a, s_km, s_ks, s_tc, b, g0, slot, tiles_per_block, bx);
}

// ---- Domain Adapter -----------------------------------------------------------------------------
// grid = B; h1 = relu(W1 label + b1) in LDS, z = W2 h1 + b2 in LDS, a = softmax over heads per ch.  Round 5: 1024 threads, a row of W2 is walked by G = 1024 / C
// threads (a contiguous run of quads each, all of its loads in flight at once), the G partial sums are added in segment order through LDS -- one thread per row was
// hid / 4 dependent L2 round trips (13-15 us per launch, 16 launches on the single-stream forward of a bs=4 step).  The result depends on (C, hid) only, never on
// the batch: the domain-batched forward stays the per-domain forwards bit for bit.
__device__ __forceinline__ void da_fwd_body(float* sm, const float* __restrict__ label, const float* __restrict__ W1, const float* __restrict__ b1,
                                            const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ a, int D, int hid, int C, int heads) {
    float* h1 = sm;                 // h1[hid], z[C], zpart[G][C]
    float* z = sm + hid;
    float* zpart = z + C;
    const int b = blockIdx.x, Ch = C / heads, BD = blockDim.x;
    for (int i = threadIdx.x; i < hid; i += BD) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(label[(long)b * D + d], W1[(long)i * D + d], s);
        h1[i] = fmaxf(s + b1[i], 0.f);
    }
    __syncthreads();
    const bool vec = (hid & 3) == 0 && (reinterpret_cast<uintptr_t>(W2) & 15) == 0;
    const int nq = vec ? hid >> 2 : hid;                       // walk units per row: quads, or single elements
    int G = BD / C;
    if (G < 1) G = 1;
    if (G > nq) G = nq;
    for (int c0 = 0; c0 < C; c0 += BD / G) {
        const int c = c0 + (int)threadIdx.x / G, seg = (int)threadIdx.x % G;
        if (c < C && (int)threadIdx.x < (BD / G) * G) {
            const int u0 = (int)((long)nq * seg / G), u1 = (int)((long)nq * (seg + 1) / G);
            const float* wr = W2 + (long)c * hid;
            float s = 0.f;
            if (vec) {
                int u = u0;
                for (; u + 8 <= u1; u += 8) {
                    float4 w4[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) w4[q] = *reinterpret_cast<const float4*>(wr + 4 * (u + q));
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int i = 4 * (u + q);
                        s = fmaf(h1[i], w4[q].x, s); s = fmaf(h1[i + 1], w4[q].y, s); s = fmaf(h1[i + 2], w4[q].z, s); s = fmaf(h1[i + 3], w4[q].w, s);
                    }
                }
                for (; u < u1; ++u) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wr + 4 * u);
                    const int i = 4 * u;
                    s = fmaf(h1[i], w4.x, s); s = fmaf(h1[i + 1], w4.y, s); s = fmaf(h1[i + 2], w4.z, s); s = fmaf(h1[i + 3], w4.w, s);
                }
            } else {
                for (int i = u0; i < u1; ++i) s = fmaf(h1[i], wr[i], s);
            }
            zpart[seg * C + c] = s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += BD) {
        float s = zpart[c];
        for (int g = 1; g < G; ++g) s += zpart[g * C + c];
        z[c] = s + b2[c];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += BD) {
        const int ch = c % Ch;
        float m = -INFINITY;
        for (int hh = 0; hh < heads; ++hh) m = fmaxf(m, z[hh * Ch + ch]);
        float s = 0.f;
        for (int hh = 0; hh < heads; ++hh) s += expf(z[hh * Ch + ch] - m);
        a[(long)b * C + c] = expf(z[c] - m) / s;
    }
}
__global__ __launch_bounds__(1024) void da_fwd_kernel(const float* __restrict__ label, const float* __restrict__ W1, const float* __restrict__ b1,
                                                      const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ a,
                                                      int D, int hid, int C, int heads) {
    extern __shared__ float sm[];
    da_fwd_body(sm, label, W1, b1, W2, b2, a, D, hid, C, heads);
}
// every adapter of a network for one label batch in ONE launch (grid = B x adapters): the adapters depend on the labels and their own weights only, so the model computes
// them at the top of its forward instead of one 13 us launch inside each of its 16 blocks (mdvit_da_fwd_many; MdvitBlockDesc.a_pre)
__global__ __launch_bounds__(1024) void da_fwd_many_kernel(MdvitDaMany m, const float* __restrict__ label, int D) {
    extern __shared__ float sm[];
    const int i = blockIdx.y;
    da_fwd_body(sm, label, m.W1[i], m.b1[i], m.W2[i], m.b2[i], m.a[i], D, m.hid[i], m.C[i], m.heads[i]);
}

// backward from e = a * dL/da:  dz[c] = e[c] - a[c] * sum_{heads} e[., ch]   (softmax over heads)
// stage 1 (grid = B): dz -> dzbuf [B,C], relu(h1) -> hbuf [B,hid], dh -> dhbuf [B,hid]
__device__ __forceinline__ void da_bwd_stage1_body(float* sm, const float* __restrict__ label, const float* __restrict__ W1, const float* __restrict__ b1,
                                                   const float* __restrict__ W2, const float* __restrict__ a, const float* __restrict__ e,
                                                   float* __restrict__ dzbuf, float* __restrict__ hbuf, float* __restrict__ dhbuf,
                                                   int D, int hid, int C, int heads, float scale) {
    float* dz = sm;                 // dz[C]
    const int b = blockIdx.x, Ch = C / heads;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int ch = c % Ch;
        float tot = 0.f;
        for (int hh = 0; hh < heads; ++hh) tot += e[(long)b * C + hh * Ch + ch];
        const float v = scale * (e[(long)b * C + c] - a[(long)b * C + c] * tot);
        dz[c] = v;
        dzbuf[(long)b * C + c] = v;
    }
    __syncthreads();
    // t[i] = sum_c dz[c] W2[c][i]: the C-long walk is split over blockDim/hid thread groups (a single thread per i paid C
    // dependent L2 round trips -- 30 us per launch on a B-block grid), partial sums meet in LDS in a fixed order
    float* tpart = sm + C;                 // [nseg][hid]
    const int BD = blockDim.x;
    const int nseg = hid <= BD ? BD / hid : 1;
    for (int i0 = 0; i0 < hid; i0 += BD) {
        const int i = i0 + (int)threadIdx.x % (hid <= BD ? hid : BD), seg = hid <= BD ? (int)threadIdx.x / hid : 0;
        if (i < hid && seg < nseg) {
            const int c_beg = (int)((long)C * seg / nseg), c_end = (int)((long)C * (seg + 1) / nseg);
            float t = 0.f;
#pragma unroll 4
            for (int c = c_beg; c < c_end; ++c) t = fmaf(dz[c], W2[(long)c * hid + i], t);
            tpart[seg * hid + i] = t;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hid; i += blockDim.x) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(label[(long)b * D + d], W1[(long)i * D + d], s);
        const float h = s + b1[i];
        float t = 0.f;
        for (int seg = 0; seg < nseg; ++seg) t += tpart[seg * hid + i];
        hbuf[(long)b * hid + i] = fmaxf(h, 0.f);
        dhbuf[(long)b * hid + i] = h > 0.f ? t : 0.f;
    }
}
__global__ __launch_bounds__(1024) void da_bwd_stage1_kernel(const float* __restrict__ label, const float* __restrict__ W1, const float* __restrict__ b1,
                                                            const float* __restrict__ W2, const float* __restrict__ a, const float* __restrict__ e,
                                                            float* __restrict__ dzbuf, float* __restrict__ hbuf, float* __restrict__ dhbuf,
                                                            int D, int hid, int C, int heads, float scale) {
    extern __shared__ float sm[];
    da_bwd_stage1_body(sm, label, W1, b1, W2, a, e, dzbuf, hbuf, dhbuf, D, hid, C, heads, scale);
}
// stage 2: one thread per weight element, loop over the batch (no atomics, deterministic)
__device__ __forceinline__ void da_bwd_stage2_body(const float* __restrict__ label, const float* __restrict__ dzbuf,
                                                   const float* __restrict__ hbuf, const float* __restrict__ dhbuf,
                                                   float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
                                                   int B, int D, int hid, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nW2 = (long)C * hid, nW1 = (long)hid * D;
    if (i < nW2) {
        const int c = (int)(i / hid), j = (int)(i % hid);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(dzbuf[(long)b * C + c], hbuf[(long)b * hid + j], s);
        dW2[i] = s;
    } else if (i < nW2 + nW1) {
        const long k = i - nW2;
        const int j = (int)(k / D), d = (int)(k % D);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(dhbuf[(long)b * hid + j], label[(long)b * D + d], s);
        dW1[k] = s;
    } else if (i < nW2 + nW1 + C) {
        const int c = (int)(i - nW2 - nW1);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dzbuf[(long)b * C + c];
        db2[c] = s;
    } else if (i < nW2 + nW1 + C + hid) {
        const int j = (int)(i - nW2 - nW1 - C);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dhbuf[(long)b * hid + j];
        db1[j] = s;
    }
}
__global__ __launch_bounds__(256) void da_bwd_stage2_kernel(const float* __restrict__ label, const float* __restrict__ dzbuf,
                                                            const float* __restrict__ hbuf, const float* __restrict__ dhbuf,
                                                            float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
                                                            int B, int D, int hid, int C) {
    da_bwd_stage2_body(label, dzbuf, hbuf, dhbuf, dW1, db1, dW2, db2, B, D, hid, C);
}
// every adapter of a network in TWO launches per backward sweep (grids B x adapters and weight-element blocks x adapters) instead of two launches of 8-12 us inside each
// of its 16 blocks on the sweep's data-gradient chain (mdvit_da_bwd_many): the blocks hand their e = a * dL/da out (MdvitBlockGrads.e_out), the adapters'
// parameter gradients are formed once the sweep has passed the first block.  An adapter with e[i] == NULL is skipped.  Same bodies, same bits.
__device__ __forceinline__ long da_many_ws_offset(const MdvitDaMany& m, int i, int B) {
    long off = 0;
    for (int j = 0; j < i; ++j) off += (long)B * m.C[j] + 2L * B * m.hid[j];
    return off;
}
__global__ __launch_bounds__(1024) void da_bwd_many_stage1_kernel(MdvitDaMany m, MdvitDaManyGrads g, const float* __restrict__ label, float* __restrict__ ws, int B, int D,
                                                                 float scale) {
    extern __shared__ float sm[];
    const int i = blockIdx.y;
    if (!g.e[i]) return;
    float* dzbuf = ws + da_many_ws_offset(m, i, B);
    float* hbuf = dzbuf + (long)B * m.C[i];
    float* dhbuf = hbuf + (long)B * m.hid[i];
    da_bwd_stage1_body(sm, label, m.W1[i], m.b1[i], m.W2[i], m.a[i], g.e[i], dzbuf, hbuf, dhbuf, D, m.hid[i], m.C[i], m.heads[i], scale);
}
__global__ __launch_bounds__(256) void da_bwd_many_stage2_kernel(MdvitDaMany m, MdvitDaManyGrads g, const float* __restrict__ label, const float* __restrict__ ws, int B, int D) {
    const int i = blockIdx.y;
    const int hid = m.hid[i], C = m.C[i];
    if (!g.e[i] || (long)blockIdx.x * blockDim.x >= (long)C * hid + (long)hid * D + C + hid) return;
    const float* dzbuf = ws + da_many_ws_offset(m, i, B);
    const float* hbuf = dzbuf + (long)B * C;
    const float* dhbuf = hbuf + (long)B * hid;
    da_bwd_stage2_body(label, dzbuf, hbuf, dhbuf, g.dW1[i], g.db1[i], g.dW2[i], g.db2[i], B, D, hid, C);
}

bool make_geom(FaGeom& g, int B, int H, int W, int C, int heads, int s3, int s5, int s7) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || C % heads != 0 || C % 4 != 0) return false;
    if (s3 + s5 + s7 != heads) return false;
    g.B = B; g.H = H; g.W = W; g.N = H * W; g.C = C; g.heads = heads; g.Ch = C / heads; g.s3 = s3; g.s5 = s5; g.s7 = s7;
    g.scale = 1.0f / sqrtf((float)g.Ch);
    return true;
}

// fwd: ws_m, ws_s [B,NT,C] + ws_P [B,NT,C,Ch].   bwd: dU, dVc [B,N,C] each + tcol [B,C] + dM [B,C,Ch] + ws_P [B,NT,C,Ch].
size_t fa_ws_floats(int B, int N, int C, int heads) {
    const int Ch = C / heads;
    const long NT = (N + FA_T - 1) / FA_T;
    const long fwd = (long)B * NT * C * (2 + Ch);
    // + the partial rows of the window-weight gradients: <= max(1024, C/32 * B) workgroups x 32 channels x (49 taps + bias)
    // (the three classes together: fa_conv3_wgrad_kernel -- at most C/32 + 2 channel blocks, and the tile chunks keep blocks x images x chunks < 1024)
    const long wg_rows = (long)(cdiv(C, 32) + 2) * B > 1024 ? (long)(cdiv(C, 32) + 2) * B : 1024;
    const long part = wg_rows * 32 * 50 > 1024L * B * C ? wg_rows * 32 * 50 : 1024L * B * C;       // ... or the e partial rows [B][<=1024][C]
    const long bwd = 2L * B * N * C + (long)B * C * (1 + Ch) + (long)B * NT * C * Ch + part;
    return (size_t)(fwd > bwd ? fwd : bwd);
}

// 64-token tiles a workgroup of the partial kernels walks.  A function of the IMAGE's token count alone -- never of the batch: an image's arithmetic (which
// tiles meet in which partial row, in which order) must not depend on how many images share the launch, or the domain-batched forward stops being the
// per-domain forwards bit for bit (tests/test_gpu_model.py: test_bench_step_fused_forward_equals_per_domain_at_512).
int fa_nsub(int NT, int /*groups*/, int /*B*/) {
    static const int env = [] { const char* e = getenv("MDVIT_FA_NSUB"); return e ? atoi(e) : 0; }();          // (experiments only: a fixed number of tiles per workgroup)
    if (env > 0) return env < NT ? env : NT;
    return NT >= 128 ? 8 : (NT >= 64 ? 4 : (NT >= 16 ? 2 : 1));
}

int quad_grid(long work_quads, int QC, int max_blocks) {
    // smallest grid >= wanted with (grid*256) % QC == 0
    int a = QC, b = 256;
    while (b) { int t = a % b; a = b; b = t; }
    const int gmul = QC / a;
    long want = (work_quads + 256L * 4 - 1) / (256L * 4);
    if (want > max_blocks) want = max_blocks;
    if (want < 1) want = 1;
    return (int)((want + gmul - 1) / gmul * gmul);
}

}  // namespace

// fa_bwd_apply_kernel's MODE (tuning hook: mdvit_factoratt_config).  Measured at 32 images (tools/probe/attn_kernel_trace.sh, profiles/r05_attn_kernels_isolated.txt): Ch = 8 / 16
// 269 / 145 us in mode 0, 460 / 240 us in mode 1 (60 registers spill), 386 / 213 us in mode 2 (4 waves per CU): a wave of these launches owns 2 tiles (the grid is kept >= 2048
// workgroups), nothing to prefetch across; the row-ahead order pays in fa_bwd_apply3_kernel, whose waves walk 4 tiles (Ch = 64: 79 -> 43 us).
int g_fa_apply_mode = 0;
int g_fa_apply_tiles = 0;       // 32-token tiles per workgroup of the apply kernels (0: the launcher's rule)
extern "C" int mdvit_factoratt_config(int32_t apply_mode, int32_t apply_tiles) {
    if (apply_mode < 0 || apply_mode > 2 || apply_tiles < 0) return mdvit_set_error(MDVIT_E_SHAPE, "factoratt_config: apply_mode in 0..2, apply_tiles >= 0");
    g_fa_apply_mode = apply_mode; g_fa_apply_tiles = apply_tiles;
    return MDVIT_OK;
}

extern "C" size_t mdvit_factoratt_ws_bytes(int32_t B, int32_t N, int32_t C, int32_t heads) {
    if (B <= 0 || N <= 0 || C <= 0 || heads <= 0 || C % heads) return 0;
    return fa_ws_floats(B, N, C, heads) * sizeof(float);
}

extern "C" int mdvit_factoratt_fwd(const float* qkv, const float* w3, const float* b3, const float* w5, const float* b5,
                                   const float* w7, const float* b7, const float* a, float* out, float* U, float* kmax, float* ksum, float* Mmat,
                                   void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                                   int32_t s3, int32_t s5, int32_t s7, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    FaGeom g;
    MDVIT_CHECK_ARG(make_geom(g, B, H, W, C, heads, s3, s5, s7), MDVIT_E_SHAPE, "factoratt_fwd: bad geometry B=%d H=%d W=%d C=%d heads=%d splits=%d/%d/%d", B, H, W, C, heads, s3, s5, s7);
    MDVIT_CHECK_ARG(ws_bytes >= fa_ws_floats(B, g.N, C, heads) * sizeof(float), MDVIT_E_WORKSPACE, "factoratt_fwd: workspace too small (%zu bytes)", ws_bytes);
    const int NT0 = cdiv(g.N, FA_T);
    const int NSUB = fa_nsub(NT0, C / (g.Ch < 32 ? 32 : g.Ch), B);
    const int NT = cdiv(NT0, NSUB);                    // partial rows per image
    float* ws_m = (float*)ws;
    float* ws_s = ws_m + (long)B * NT * C;
    float* ws_P = ws_s + (long)B * NT * C;
    {
        const int GW = g.Ch < 32 ? 32 : g.Ch;
        MDVIT_CHECK_ARG(C % GW == 0, MDVIT_E_SHAPE, "factoratt_fwd: C=%d is not a multiple of the %d-channel group", C, GW);
#define FA_PART_LAUNCH(CHV) hipLaunchKernelGGL((fa_partial_kernel<CHV, true>), dim3(NT, C / GW, B), dim3(256), 0, s, \
                       qkv + C, (long)3 * C, qkv + 2 * C, (long)3 * C, (const float*)nullptr, 1.f, ws_m, ws_s, ws_P, g, NT0, NSUB)
        static const bool s8_env = [] { const char* e = getenv("MDVIT_FA_PARTIAL_STREAM"); return !(e && e[0] == '0'); }();
        static const bool s16_env = [] { const char* e = getenv("MDVIT_FA_PARTIAL_STREAM16"); return !(e && e[0] == '0'); }();
        if (g.Ch == 8 && C == 64 && s8_env)          // the streaming form (MDVIT_FA_PARTIAL_STREAM=0: the LDS / MFMA tiles, A/B)
            fa_partial_stream_launch<8, true>(dim3(NT, 1, B), s, qkv + C, (long)3 * C, qkv + 2 * C, (long)3 * C, nullptr, 1.f, ws_m, ws_s, ws_P, g, NT0, NSUB, nullptr, nullptr, nullptr);
        else if (g.Ch == 16 && C == 128 && s8_env && s16_env)
            fa_partial_stream_launch<16, true>(dim3(NT, 1, B), s, qkv + C, (long)3 * C, qkv + 2 * C, (long)3 * C, nullptr, 1.f, ws_m, ws_s, ws_P, g, NT0, NSUB, nullptr, nullptr, nullptr);
        else
        switch (g.Ch) {
            case 8: FA_PART_LAUNCH(8); break;
            case 16: FA_PART_LAUNCH(16); break;
            case 40: FA_PART_LAUNCH(40); break;
            case 64: FA_PART_LAUNCH(64); break;
            default: return mdvit_set_error(MDVIT_E_SHAPE, "factoratt_fwd: head dim %d not built (8/16/40/64)", g.Ch);
        }
#undef FA_PART_LAUNCH
    }
    if (NT > 8)
        hipLaunchKernelGGL((fa_combine_softmax_kernel<32>), dim3(cdiv((long)C * g.Ch, 8), B), dim3(256), 0, s, ws_m, ws_s, ws_P, kmax, ksum, Mmat, g, NT);
    else if (NT > 2)
        hipLaunchKernelGGL((fa_combine_softmax_kernel<4>), dim3(cdiv((long)C * g.Ch, 64), B), dim3(256), 0, s, ws_m, ws_s, ws_P, kmax, ksum, Mmat, g, NT);
    else
        hipLaunchKernelGGL((fa_combine_softmax_kernel<1>), dim3(cdiv((long)C * g.Ch, 256), B), dim3(256), 0, s, ws_m, ws_s, ws_P, kmax, ksum, Mmat, g, NT);
    MDVIT_CHECK_ARG(C <= 512, MDVIT_E_SHAPE, "factoratt_fwd: C=%d > 512 not built", C);
    // U = dwconv_win(v) + bias, one tiled launch per window class (channels [0,s3*Ch) | [..) | [..))
    const int Ch = g.Ch, c5 = s3 * Ch, c7 = (s3 + s5) * Ch;
    const CtGeom cg{B, g.H, g.W};
    {   // the three window classes in one launch
        const int xoff[3] = {2 * C, 2 * C + c5, 2 * C + c7}, yoff[3] = {0, c5, c7}, ncls[3] = {s3 * Ch, s5 * Ch, s7 * Ch};
        const float* const ws3[3] = {w3, w5, w7};
        const float* const bs3[3] = {b3, b5, b7};
        launch_conv3<false>(qkv, 3L * C, xoff, ws3, bs3, U, (long)C, yoff, cg, ncls, s);
    }
    {
        const int TLN = max(1, 256 / C), block = TLN * C;
        int tpb = TLN * 32;                       // >= 32 tokens per thread: the CH-register column preload is amortised
        while (tpb > TLN * 8 && (long)cdiv(g.N, tpb) * B < 1024) tpb /= 2;
        while ((long)cdiv(g.N, tpb) * B > 8192) tpb *= 2;
        dim3 grid(cdiv(g.N, tpb), B);
#define FA_OUT_LAUNCH(CHV) hipLaunchKernelGGL((fa_out_kernel<CHV>), grid, dim3(block), sizeof(float) * FA_OUT_U * TLN * C, s, qkv, U, Mmat, a, out, g, TLN, tpb)
        switch (Ch) {
            case 8: FA_OUT_LAUNCH(8); break;
            case 16: FA_OUT_LAUNCH(16); break;
            case 40: FA_OUT_LAUNCH(40); break;
            case 64: FA_OUT_LAUNCH(64); break;
            default: return mdvit_set_error(MDVIT_E_SHAPE, "factoratt_fwd: head dim %d not built (8/16/40/64)", Ch);
        }
#undef FA_OUT_LAUNCH
    }
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_factoratt_bwd(const float* dout, const float* qkv, const float* out, const float* U,
                                   const float* w3, const float* b3, const float* w5, const float* b5, const float* w7, const float* b7,
                                   const float* a, const float* kmax, const float* ksum, const float* Mmat,
                                   float* dqkv, float* e, float* dw3, float* db3, float* dw5, float* db5, float* dw7, float* db7,
                                   void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                                   int32_t s3, int32_t s5, int32_t s7, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    FaGeom g;
    MDVIT_CHECK_ARG(make_geom(g, B, H, W, C, heads, s3, s5, s7), MDVIT_E_SHAPE, "factoratt_bwd: bad geometry B=%d H=%d W=%d C=%d heads=%d", B, H, W, C, heads);
    MDVIT_CHECK_ARG(ws_bytes >= fa_ws_floats(B, g.N, C, heads) * sizeof(float), MDVIT_E_WORKSPACE, "factoratt_bwd: workspace too small (%zu bytes)", ws_bytes);
    MDVIT_CHECK_ARG(C <= 512, MDVIT_E_SHAPE, "factoratt_bwd: C=%d > 512 not built", C);
    MDVIT_CHECK_ARG((a == nullptr) == (e == nullptr), MDVIT_E_SHAPE, "factoratt_bwd: a and e must both be given or both be NULL");
    const int Ch = g.Ch, NT = cdiv(g.N, FA_T);
    float* dU = (float*)ws;
    float* dVc = dU + (long)B * g.N * C;
    float* tcol = dVc + (long)B * g.N * C;
    float* dM = tcol + (long)B * C;
    float* ws_P = dM + (long)B * C * Ch;
    const bool want_wgrad = dw3 != nullptr;
    MDVIT_CHECK_ARG(want_wgrad ? (db3 && dw5 && db5 && dw7 && db7) : !(db3 || dw5 || db5 || dw7 || db7), MDVIT_E_SHAPE,
                    "factoratt_bwd: the six crpe gradient outputs must be all given or all NULL");
    if (dqkv == nullptr) {
        // the adapter's gradient carrier only: e = sum_n dout * out (the data-gradient-only aux sweep at the FIRST adapter of the network --
        // nothing below it carries an adapter, so nothing below it is needed)
        MDVIT_CHECK_ARG(e != nullptr && !want_wgrad, MDVIT_E_SHAPE, "factoratt_bwd: dqkv == NULL asks for e alone");
        const int QC0 = C / 4;
        const int gx = quad_grid((long)g.N * QC0, QC0, 512);
        float* e_part = ws_P + (long)B * NT * C * Ch;
        hipLaunchKernelGGL(fa_bwd_prep_kernel, dim3(gx, B), dim3(256), sizeof(float) * C, s, dout, qkv, out, a, (float*)nullptr, e_part, g);
        MDVIT_LAUNCH_CHECK();
        return mdvit_reduce_partials_batched(e_part, B, gx, C, e, s);
    }
    // 1 + 2: dM = Q^T (scale * a * G) as tile partials, and on the same staged tiles dU = a G q and the e partial rows
    const int GWp = Ch < 32 ? 32 : Ch;
    MDVIT_CHECK_ARG(C % GWp == 0, MDVIT_E_SHAPE, "factoratt_bwd: C=%d is not a multiple of the %d-channel group", C, GWp);
    const int NSUBb = fa_nsub(NT, C / GWp, B), NTS = cdiv(NT, NSUBb);
    float* e_part = ws_P + (long)B * NT * C * Ch;               // [B][NTS][C] partial rows (the region is reused by the window-weight gradients below)
    {
#define FA_PART_LAUNCH(CHV) hipLaunchKernelGGL((fa_partial_kernel<CHV, false>), dim3(NTS, C / GWp, B), dim3(256), 0, s, \
                       qkv, (long)3 * C, dout, (long)C, a, g.scale, (float*)nullptr, (float*)nullptr, ws_P, g, NT, NSUBb, out, dU, e ? e_part : (float*)nullptr)
        static const bool s8p_env = [] { const char* e = getenv("MDVIT_FA_PARTIAL_STREAM"); return !(e && e[0] == '0'); }();
        static const bool s16p_env = [] { const char* e = getenv("MDVIT_FA_PARTIAL_STREAM16"); return !(e && e[0] == '0'); }();
        if (Ch == 8 && C == 64 && s8p_env)
            fa_partial_stream_launch<8, false>(dim3(NTS, 1, B), s, qkv, (long)3 * C, dout, (long)C, a, g.scale, nullptr, nullptr, ws_P, g, NT, NSUBb, out, dU, e ? e_part : nullptr);
        else if (Ch == 16 && C == 128 && s8p_env && s16p_env)
            fa_partial_stream_launch<16, false>(dim3(NTS, 1, B), s, qkv, (long)3 * C, dout, (long)C, a, g.scale, nullptr, nullptr, ws_P, g, NT, NSUBb, out, dU, e ? e_part : nullptr);
        else
        switch (Ch) {
            case 8: FA_PART_LAUNCH(8); break;
            case 16: FA_PART_LAUNCH(16); break;
            case 40: FA_PART_LAUNCH(40); break;
            case 64: FA_PART_LAUNCH(64); break;
            default: return mdvit_set_error(MDVIT_E_SHAPE, "factoratt_bwd: head dim %d not built (8/16/40/64)", Ch);
        }
#undef FA_PART_LAUNCH
    }
    if (e) {
        const int rc = mdvit_reduce_partials_batched(e_part, B, NTS, C, e, s);
        if (rc != MDVIT_OK) return rc;
    }
    // 3: crpe weight gradients
    const int c5 = s3 * Ch, c7 = (s3 + s5) * Ch;
    const CtGeom cg{B, g.H, g.W};
    if (want_wgrad) {
        float* wg_part = ws_P + (long)B * NT * C * Ch;          // partial rows, reduced by the finish kernel of each class
        const int goff[3] = {0, c5, c7}, xoff3[3] = {2 * C, 2 * C + c5, 2 * C + c7}, ncls3[3] = {s3 * Ch, s5 * Ch, s7 * Ch};
        float* const dws[3] = {dw3, dw5, dw7};
        float* const dbs[3] = {db3, db5, db7};
        launch_conv3_wgrad(dU, (long)C, goff, qkv, 3L * C, xoff3, dws, dbs, wg_part, cg, ncls3, s, 0);
    }
    // conv^T(dU) = correlation with the flipped window
    {   // transposed (flipped-tap) windows of the three classes in one launch
        const int off[3] = {0, c5, c7}, ncls[3] = {s3 * Ch, s5 * Ch, s7 * Ch};
        const float* const ws3[3] = {w3, w5, w7};
        launch_conv3<true>(dU, (long)C, off, ws3, nullptr, dVc, (long)C, off, cg, ncls, s);
    }
    // 4, 5
    const int GW = Ch < 32 ? 32 : Ch;
    MDVIT_CHECK_ARG(C % GW == 0, MDVIT_E_SHAPE, "factoratt_bwd: C=%d is not a multiple of the %d-channel group", C, GW);
    const int ntiles = cdiv(g.N, 32);
    if (Ch >= 32) {                                // products dealt to wavefronts: 2 tiles x 3 roles per workgroup pass
        // 8 tiles per workgroup (4 per wave slot) amortise the staging of the two Ch x Ch matrices; fewer only while the grid would not reach one workgroup per CU
        // (measured at 32 images, C = 320 / 512: 2 tiles 227 / 157 us, 4: 163 / 97, 8: 144 / 71, 32: 140 / 71)
        int tpb = 8;
        while (tpb > 2 && (long)cdiv(ntiles, tpb) * (C / GW) * B < 256) tpb /= 2;
        if (g_fa_apply_tiles > 0) tpb = g_fa_apply_tiles;
        const int gx = cdiv(ntiles, tpb);
        const long nwg = (long)gx * (C / GW) * B;
        static const bool xcd_env = [] { const char* e = getenv("MDVIT_FA_APPLY3_XCD"); return !(e && e[0] == '0'); }();
        const int xcd_map = xcd_env && nwg % 8 == 0 && nwg < (1L << 30);          // (see the kernel: the heads of a token range on one XCD)
        dim3 grid = xcd_map ? dim3((unsigned)nwg, 1, 1) : dim3(gx, C / GW, B);
        if (Ch == 40) hipLaunchKernelGGL((fa_bwd_apply3_kernel<40>), grid, dim3(384), 0, s, dout, qkv, U, dVc, Mmat, a, kmax, ksum, ws_P, NTS, dqkv, g, tpb, gx, xcd_map);
        else if (Ch == 64) hipLaunchKernelGGL((fa_bwd_apply3_kernel<64>), grid, dim3(384), 0, s, dout, qkv, U, dVc, Mmat, a, kmax, ksum, ws_P, NTS, dqkv, g, tpb, gx, xcd_map);
        else return mdvit_set_error(MDVIT_E_SHAPE, "factoratt_bwd: head dim %d not built (8/16/40/64)", Ch);
        MDVIT_LAUNCH_CHECK();
        return MDVIT_OK;
    }
    static const bool s8_env = [] { const char* e = getenv("MDVIT_FA_APPLY_STREAM"); return !(e && e[0] == '0'); }();
    if (Ch == 8 && C == 64 && s8_env && g_fa_apply_mode == 0) {          // the streaming VALU form (MDVIT_FA_APPLY_STREAM=0: the MFMA tiles, A/B)
        int tpbk = 1024;                                              // tokens per workgroup: halved while the launch has fewer than two workgroups per CU
        while (tpbk > 128 && (long)cdiv(g.N, tpbk) * B < 512) tpbk /= 2;
        if (g_fa_apply_tiles > 0) tpbk = 32 * g_fa_apply_tiles;
        // the Ch = 8 matrices in the LDS table too (214.8 against 221.7 us at 32 images with them in 96 registers; MDVIT_FA_APPLY_S8_TABLE=0: the register form, A/B)
        static const bool s8_tab = [] { const char* e = getenv("MDVIT_FA_APPLY_S8_TABLE"); return !(e && e[0] == '0'); }();
        if (s8_tab) hipLaunchKernelGGL(fa_bwd_apply_tab_kernel<8>, dim3(cdiv(g.N, tpbk), B), dim3(256), 0, s, dout, qkv, U, dVc, Mmat, a, kmax, ksum, ws_P, NTS, dqkv, g, tpbk);
        else hipLaunchKernelGGL(fa_bwd_apply_s8_kernel, dim3(cdiv(g.N, tpbk), B), dim3(256), 0, s, dout, qkv, U, dVc, Mmat, a, kmax, ksum, ws_P, NTS, dqkv, g, tpbk);
        MDVIT_LAUNCH_CHECK();
        return MDVIT_OK;
    }
    static const bool s16_env = [] { const char* e = getenv("MDVIT_FA_APPLY_STREAM16"); return !(e && e[0] == '0'); }();
    if (Ch == 16 && C == 128 && s8_env && s16_env && g_fa_apply_mode == 0) {
        int tpbk = 512;
        while (tpbk > 128 && (long)cdiv(g.N, tpbk) * B < 512) tpbk /= 2;
        if (g_fa_apply_tiles > 0) tpbk = 32 * g_fa_apply_tiles;
        hipLaunchKernelGGL(fa_bwd_apply_tab_kernel<16>, dim3(cdiv(g.N, tpbk), B), dim3(256), 0, s, dout, qkv, U, dVc, Mmat, a, kmax, ksum, ws_P, NTS, dqkv, g, tpbk);
        MDVIT_LAUNCH_CHECK();
        return MDVIT_OK;
    }
    int tpb = 4;                                   // 32-token tiles per block (one per wavefront), doubled while the grid stays large
    while (tpb < 64 && (long)cdiv(ntiles, tpb * 2) * (C / GW) * B >= 2048) tpb *= 2;
    if (g_fa_apply_tiles > 0) tpb = g_fa_apply_tiles;
    dim3 grid(cdiv(ntiles, tpb), C / GW, B);
#define FA_BWD_LAUNCH(CHV, MODEV) hipLaunchKernelGGL((fa_bwd_apply_kernel<CHV, MODEV>), grid, dim3(256), 0, s, dout, qkv, U, dVc, Mmat, a, kmax, ksum, ws_P, NTS, dqkv, g, tpb)
    switch (Ch * 4 + g_fa_apply_mode) {
        case 8 * 4 + 0: FA_BWD_LAUNCH(8, 0); break;
        case 8 * 4 + 1: FA_BWD_LAUNCH(8, 1); break;
        case 8 * 4 + 2: FA_BWD_LAUNCH(8, 2); break;
        case 16 * 4 + 0: FA_BWD_LAUNCH(16, 0); break;
        case 16 * 4 + 1: FA_BWD_LAUNCH(16, 1); break;
        case 16 * 4 + 2: FA_BWD_LAUNCH(16, 2); break;
        default: return mdvit_set_error(MDVIT_E_SHAPE, "factoratt_bwd: head dim %d not built (8/16/40/64)", Ch);
    }
#undef FA_BWD_LAUNCH
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

/* The window-weight gradients of a preceding mdvit_factoratt_bwd call (same geometry, same `ws`, given dw3 == NULL there):
 * they only read dU (left in ws) and v, so they may run on ANOTHER stream once the backward's kernels are ordered before
 * it; accumulate != 0 adds into dw/db (gradient buckets). */
extern "C" int mdvit_factoratt_wgrad(const float* qkv, void* ws, size_t ws_bytes, float* dw3, float* db3, float* dw5, float* db5,
                                     float* dw7, float* db7, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                                     int32_t s3, int32_t s5, int32_t s7, int32_t accumulate, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    FaGeom g;
    MDVIT_CHECK_ARG(make_geom(g, B, H, W, C, heads, s3, s5, s7), MDVIT_E_SHAPE, "factoratt_wgrad: bad geometry");
    MDVIT_CHECK_ARG(ws_bytes >= fa_ws_floats(B, g.N, C, heads) * sizeof(float), MDVIT_E_WORKSPACE, "factoratt_wgrad: workspace too small");
    MDVIT_CHECK_ARG(dw3 && db3 && dw5 && db5 && dw7 && db7, MDVIT_E_SHAPE, "factoratt_wgrad: all six outputs are required");
    const int Ch = g.Ch, NT = cdiv(g.N, FA_T);
    const float* dU = (const float*)ws;
    float* wg_part = (float*)ws + 2L * B * g.N * C + (long)B * C * (1 + Ch) + (long)B * NT * C * Ch;
    const int c5 = s3 * Ch, c7 = (s3 + s5) * Ch;
    const CtGeom cg{B, g.H, g.W};
    const int goff[3] = {0, c5, c7}, xoff3[3] = {2 * C, 2 * C + c5, 2 * C + c7}, ncls3[3] = {s3 * Ch, s5 * Ch, s7 * Ch};
    float* const dws[3] = {dw3, dw5, dw7};
    float* const dbs[3] = {db3, db5, db7};
    launch_conv3_wgrad(dU, (long)C, goff, qkv, 3L * C, xoff3, dws, dbs, wg_part, cg, ncls3, s, accumulate);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_da_fwd(const float* label, const float* W1, const float* b1, const float* W2, const float* b2, float* a,
                            int32_t B, int32_t D, int32_t hid, int32_t C, int32_t heads, void* stream) {
    MDVIT_CHECK_ARG(B > 0 && D > 0 && hid > 0 && C > 0 && heads > 0 && C % heads == 0, MDVIT_E_SHAPE, "da_fwd: bad shape");
    const int G = max(1, min(1024 / C, hid));            // (an upper bound of the kernel's G: it walks quads when it can)
    hipLaunchKernelGGL(da_fwd_kernel, dim3(B), dim3(1024), sizeof(float) * (hid + C + (size_t)(G + 1) * C), (hipStream_t)stream, label, W1, b1, W2, b2, a, D, hid, C, heads);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_da_fwd_many(const MdvitDaMany* m, const float* label, int32_t B, int32_t D, void* stream) {
    MDVIT_CHECK_ARG(m && label && B > 0 && D > 0 && m->n > 0 && m->n <= MDVIT_DA_MANY_MAX, MDVIT_E_SHAPE, "da_fwd_many: bad arguments");
    size_t smem = 0;
    for (int i = 0; i < m->n; ++i) {
        const int C = m->C[i], hid = m->hid[i];
        MDVIT_CHECK_ARG(m->W1[i] && m->b1[i] && m->W2[i] && m->b2[i] && m->a[i] && hid > 0 && C > 0 && m->heads[i] > 0 && C % m->heads[i] == 0, MDVIT_E_SHAPE,
                        "da_fwd_many: bad adapter %d", i);
        const int G = max(1, min(1024 / C, hid));
        smem = std::max(smem, sizeof(float) * (hid + C + (size_t)(G + 1) * C));
    }
    MDVIT_CHECK_ARG(smem <= 64 * 1024, MDVIT_E_SHAPE, "da_fwd_many: an adapter needs %zu bytes of LDS", smem);
    hipLaunchKernelGGL(da_fwd_many_kernel, dim3(B, m->n), dim3(1024), smem, (hipStream_t)stream, *m, label, D);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" size_t mdvit_da_ws_bytes(int32_t B, int32_t hid, int32_t C) {
    if (B <= 0 || hid <= 0 || C <= 0) return 0;
    return sizeof(float) * ((size_t)B * C + 2 * (size_t)B * hid);
}

extern "C" size_t mdvit_da_many_ws_bytes(const MdvitDaMany* m, int32_t B) {
    if (!m || m->n <= 0 || m->n > MDVIT_DA_MANY_MAX || B <= 0) return 0;
    size_t t = 0;
    for (int i = 0; i < m->n; ++i) t += mdvit_da_ws_bytes(B, m->hid[i], m->C[i]);
    return t;
}

extern "C" int mdvit_da_bwd_many(const MdvitDaMany* m, const MdvitDaManyGrads* g, const float* label, float scale, void* ws, size_t ws_bytes, int32_t B, int32_t D,
                                 void* stream) {
    MDVIT_CHECK_ARG(m && g && m->n > 0 && m->n <= MDVIT_DA_MANY_MAX && B > 0 && D > 0, MDVIT_E_SHAPE, "da_bwd_many: bad shape");
    MDVIT_CHECK_ARG(ws_bytes >= mdvit_da_many_ws_bytes(m, B), MDVIT_E_WORKSPACE, "da_bwd_many: workspace too small");
    size_t smem = 0;
    long most = 0;
    bool any = false;
    for (int i = 0; i < m->n; ++i) {
        MDVIT_CHECK_ARG(m->hid[i] > 0 && m->C[i] > 0 && m->heads[i] > 0 && m->C[i] % m->heads[i] == 0, MDVIT_E_SHAPE, "da_bwd_many: bad adapter shape");
        if (!g->e[i]) continue;
        MDVIT_CHECK_ARG(g->dW1[i] && g->db1[i] && g->dW2[i] && g->db2[i], MDVIT_E_SHAPE, "da_bwd_many: an adapter with e needs its four gradient outputs");
        any = true;
        smem = max(smem, sizeof(float) * (m->C[i] + (size_t)max(m->hid[i], 1024)));
        most = max(most, (long)m->C[i] * m->hid[i] + (long)m->hid[i] * D + m->C[i] + m->hid[i]);
    }
    if (!any) return MDVIT_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(da_bwd_many_stage1_kernel, dim3(B, m->n), dim3(1024), smem, s, *m, *g, label, (float*)ws, B, D, scale);
    hipLaunchKernelGGL(da_bwd_many_stage2_kernel, dim3(cdiv(most, 256), m->n), dim3(256), 0, s, *m, *g, label, (const float*)ws, B, D);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_da_bwd(const float* label, const float* W1, const float* b1, const float* W2, const float* b2, const float* a,
                            const float* e, float scale, float* dW1, float* db1, float* dW2, float* db2, void* ws, size_t ws_bytes,
                            int32_t B, int32_t D, int32_t hid, int32_t C, int32_t heads, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    (void)b2;
    MDVIT_CHECK_ARG(B > 0 && D > 0 && hid > 0 && C > 0 && heads > 0 && C % heads == 0, MDVIT_E_SHAPE, "da_bwd: bad shape");
    MDVIT_CHECK_ARG(ws_bytes >= mdvit_da_ws_bytes(B, hid, C), MDVIT_E_WORKSPACE, "da_bwd: workspace too small");
    float* dzbuf = (float*)ws;
    float* hbuf = dzbuf + (long)B * C;
    float* dhbuf = hbuf + (long)B * hid;
    hipLaunchKernelGGL(da_bwd_stage1_kernel, dim3(B), dim3(1024), sizeof(float) * (C + (size_t)max(hid, 1024)), s, label, W1, b1, W2, a, e, dzbuf, hbuf, dhbuf, D, hid, C, heads, scale);
    const long total = (long)C * hid + (long)hid * D + C + hid;
    hipLaunchKernelGGL(da_bwd_stage2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, label, dzbuf, hbuf, dhbuf, dW1, db1, dW2, db2, B, D, hid, C);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
