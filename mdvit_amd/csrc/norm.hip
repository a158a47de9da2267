// LayerNorm, BatchNorm(+activation, +Dropout2d), 1-output linear ("rowdot") and column sums.
// All HBM-bound streaming kernels over token-major [M, C] fp32: lanes run along C (coalesced),
// row statistics by wavefront shuffle reductions, channel statistics / parameter gradients by per-thread partials ->
// LDS -> one partial row per workgroup -> fixed-order second stage (no global atomics in this file).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per token row, row held in registers (C <= 64*VPT).
// ------------------------------------------------------------------------------------------------
// grid.y = parameter group: `M` consecutive rows per group, each with its own gamma/beta row (one launch normalises a
// domain-batched tensor with per-domain LayerNorm parameters; groups = 1 is the plain LayerNorm)
#define LN_GROUP_FWD(C_)                                                                                   \
    { const long g_ = blockIdx.y; x += g_ * M * (C_); y += g_ * M * (C_); mean += g_ * M; rstd += g_ * M;   \
      gamma += g_ * (C_); beta += g_ * (C_); }
#define LN_GROUP_BWD(C_)                                                                                   \
    { const long g_ = blockIdx.y; dy += g_ * M * (C_); x += g_ * M * (C_); dx += g_ * M * (C_);             \
      if (add) add += g_ * M * (C_);                                                                       \
      mean += g_ * M; rstd += g_ * M; gamma += g_ * (C_); part += g_ * gridDim.x * 2 * (C_); }

template <int VPT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int C, float eps) {
    LN_GROUP_FWD(C);
    const int lane = threadIdx.x & 63;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    for (int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < M; row += nw) {
        const float* xr = x + (long)row * C;
        float v[VPT];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            v[j] = c < C ? xr[c] : 0.f;
            s += v[j];
        }
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            const float d = c < C ? v[j] - mu : 0.f;
            q += d * d;
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
        float* yr = y + (long)row * C;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            if (c < C) yr[c] = (v[j] - mu) * rs * gamma[c] + beta[c];
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

template <int VPT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ add, float* __restrict__ dx,
                                                     float* __restrict__ part, int M, int C) {
    LN_GROUP_BWD(C);
    __shared__ float s_dg[4][64 * VPT];
    __shared__ float s_db[4][64 * VPT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    float adg[VPT], adb[VPT], g[VPT];
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        adg[j] = 0.f; adb[j] = 0.f;
        const int c = lane + 64 * j;
        g[j] = c < C ? gamma[c] : 0.f;
    }
    for (int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < M; row += nw) {
        const float mu = mean[row], rs = rstd[row];
        float xh[VPT], d[VPT];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            const bool ok = c < C;
            const float dv = ok ? dy[(long)row * C + c] : 0.f;
            xh[j] = ok ? (x[(long)row * C + c] - mu) * rs : 0.f;
            d[j] = dv * g[j];
            c1 += d[j];
            c2 += d[j] * xh[j];
            adg[j] += dv * xh[j];
            adb[j] += dv;
        }
        c1 = wave_sum(c1) / (float)C;
        c2 = wave_sum(c2) / (float)C;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int c = lane + 64 * j;
            if (c < C) dx[(long)row * C + c] = rs * (d[j] - c1 - xh[j] * c2) + (add ? add[(long)row * C + c] : 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) { s_dg[wave][lane + 64 * j] = adg[j]; s_db[wave][lane + 64 * j] = adb[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        part[(long)blockIdx.x * 2 * C + c] = s_dg[0][c] + s_dg[1][c] + s_dg[2][c] + s_dg[3][c];           // [dgamma | dbeta] row
        part[(long)blockIdx.x * 2 * C + C + c] = s_db[0][c] + s_db[1][c] + s_db[2][c] + s_db[3][c];
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm for C = 64*VPL (the model's 64/128/320/512): 16 lanes per token row (4 rows per wavefront), each lane
// VPL float4 quads strided by 16 -> 16-byte coalesced loads, 4-step 16-lane shuffle reductions, 4 independent rows
// in flight per wavefront (the wave-per-row kernels above move 4 bytes per lane and serialise on 6-step reductions).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    return v;
}

// (ln_sq4 / ln_affine4 of common.h: explicitly rounded, so that the LayerNorm prologue of mdvit_linear_rc_ln reproduces this kernel bit for bit)
template <int VPL>
__global__ __launch_bounds__(256) void ln_fwd16_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int M, float eps) {
    constexpr int C = 64 * VPL;
    LN_GROUP_FWD(C);
    const int sub = threadIdx.x & 15;
    const long r0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4, nr = ((long)gridDim.x * blockDim.x) >> 4;
    float4 ga[VPL], be[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        ga[j] = *reinterpret_cast<const float4*>(gamma + 4 * (sub + 16 * j));
        be[j] = *reinterpret_cast<const float4*>(beta + 4 * (sub + 16 * j));
    }
    for (long row = r0; row < M; row += nr) {
        const float* xr = x + row * C;
        float4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) { v[j] = *reinterpret_cast<const float4*>(xr + 4 * (sub + 16 * j)); s += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
        const float mu = sum16(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
            q += ln_sq4(v[j]);
        }
        const float rs = mdvit_ln_rstd(sum16(q), 1.0f / C, eps);
        float* yr = y + row * C;
#pragma unroll
        for (int j = 0; j < VPL; ++j) *reinterpret_cast<float4*>(yr + 4 * (sub + 16 * j)) = ln_affine4(v[j], rs, ga[j], be[j]);
        if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// optional second output of the LayerNorm backward: dx * dropout mask * DropPath row scale of the Linear that produced the LayerNorm's input
struct LnMasked {
    float* out; float drop_p; uint32_t k0, k1, thresh; float inv_keep; const float* rowscale; int rows_per_scale; const uint32_t* seed;
    const float* dy2;       // a second partial of the upstream gradient, added to dy on the fly (mdvit_mlp_rc_bwd writes dx as one partial per hidden role)
};

template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd16_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ add, float* __restrict__ dx,
                                                       float* __restrict__ part, int M, const LnMasked mk) {
    constexpr int C = 64 * VPL;
    const long grow0 = (long)blockIdx.y * M;         // first row of this parameter group in the whole [rows, C] tensor (mask indices, row scales)
    LN_GROUP_BWD(C);
    const uint32_t k0e = mk.k0 ^ (mk.seed ? mk.seed[0] : 0u), k1e = mk.k1 + (mk.seed ? mk.seed[1] : 0u);
    __shared__ float s_dg[4][C], s_db[4][C];
    const int sub = threadIdx.x & 15, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4, nr = ((long)gridDim.x * blockDim.x) >> 4;
    float4 ga[VPL], adg[VPL], adb[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        ga[j] = *reinterpret_cast<const float4*>(gamma + 4 * (sub + 16 * j));
        adg[j] = make_float4(0.f, 0.f, 0.f, 0.f); adb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long row = r0; row < M; row += nr) {
        const float mu = mean[row], rs = rstd[row];
        float4 xh[VPL], d[VPL];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            float4 g4 = *reinterpret_cast<const float4*>(dy + row * C + 4 * (sub + 16 * j));
            if (mk.dy2) {
                const float4 h4 = *reinterpret_cast<const float4*>(mk.dy2 + (grow0 + row) * C + 4 * (sub + 16 * j));
                g4.x += h4.x; g4.y += h4.y; g4.z += h4.z; g4.w += h4.w;
            }
            const float4 x4 = *reinterpret_cast<const float4*>(x + row * C + 4 * (sub + 16 * j));
            xh[j] = make_float4((x4.x - mu) * rs, (x4.y - mu) * rs, (x4.z - mu) * rs, (x4.w - mu) * rs);
            d[j] = make_float4(g4.x * ga[j].x, g4.y * ga[j].y, g4.z * ga[j].z, g4.w * ga[j].w);
            c1 += (d[j].x + d[j].y) + (d[j].z + d[j].w);
            c2 += (d[j].x * xh[j].x + d[j].y * xh[j].y) + (d[j].z * xh[j].z + d[j].w * xh[j].w);
            adg[j].x += g4.x * xh[j].x; adg[j].y += g4.y * xh[j].y; adg[j].z += g4.z * xh[j].z; adg[j].w += g4.w * xh[j].w;
            adb[j].x += g4.x; adb[j].y += g4.y; adb[j].z += g4.z; adb[j].w += g4.w;
        }
        c1 = sum16(c1) * (1.0f / C);
        c2 = sum16(c2) * (1.0f / C);
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            float4 o = make_float4(rs * (d[j].x - c1 - xh[j].x * c2), rs * (d[j].y - c1 - xh[j].y * c2),
                                   rs * (d[j].z - c1 - xh[j].z * c2), rs * (d[j].w - c1 - xh[j].w * c2));
            if (add) {           // gradient of the residual branch that forked off the LayerNorm input
                const float4 a4 = *reinterpret_cast<const float4*>(add + row * C + 4 * (sub + 16 * j));
                o.x += a4.x; o.y += a4.y; o.z += a4.z; o.w += a4.w;
            }
            *reinterpret_cast<float4*>(dx + row * C + 4 * (sub + 16 * j)) = o;
            if (mk.out) {        // the consumer Linear's masked upstream gradient (chan_reduce_kernel<2>'s arithmetic, no second pass over dx)
                const long gr = grow0 + row;
                const float r = mk.rowscale ? mk.rowscale[gr / mk.rows_per_scale] : 1.f;
                float4 ds = make_float4(r, r, r, r);
                if (mk.drop_p > 0.f) {
                    const float4 d4 = mdvit_drop_scale4(k0e, k1e, (uint32_t)(gr * C + 4 * (sub + 16 * j)), mk.thresh, mk.inv_keep);
                    ds.x *= d4.x; ds.y *= d4.y; ds.z *= d4.z; ds.w *= d4.w;
                }
                *reinterpret_cast<float4*>(mk.out + gr * C + 4 * (sub + 16 * j)) = make_float4(o.x * ds.x, o.y * ds.y, o.z * ds.z, o.w * ds.w);
            }
        }
    }
    if (part == nullptr) return;          // data-gradient-only sweep: no parameter gradients wanted (uniform)
    // fold the 4 row slots of the wavefront (lanes sub, sub+16, sub+32, sub+48), then the 4 wavefronts through LDS
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            adg[j].x += __shfl_xor(adg[j].x, o, 64); adg[j].y += __shfl_xor(adg[j].y, o, 64);
            adg[j].z += __shfl_xor(adg[j].z, o, 64); adg[j].w += __shfl_xor(adg[j].w, o, 64);
            adb[j].x += __shfl_xor(adb[j].x, o, 64); adb[j].y += __shfl_xor(adb[j].y, o, 64);
            adb[j].z += __shfl_xor(adb[j].z, o, 64); adb[j].w += __shfl_xor(adb[j].w, o, 64);
        }
        if (lane < 16) {
            *reinterpret_cast<float4*>(&s_dg[wave][4 * (sub + 16 * j)]) = adg[j];
            *reinterpret_cast<float4*>(&s_db[wave][4 * (sub + 16 * j)]) = adb[j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        part[(long)blockIdx.x * 2 * C + c] = (s_dg[0][c] + s_dg[1][c]) + (s_dg[2][c] + s_dg[3][c]);       // [dgamma | dbeta] row
        part[(long)blockIdx.x * 2 * C + C + c] = (s_db[0][c] + s_db[1][c]) + (s_db[2][c] + s_db[3][c]);
    }
}

// ------------------------------------------------------------------------------------------------
// Channel reductions over [M,C] viewed as float4 quads: total thread count is a multiple of C/4 so
// every thread owns ONE channel quad.  MODE 0: sum & sumsq of y (BN stats).  MODE 1: BN backward
// sums (sum g, sum g*xhat) with g = dz * drop2d * act'(pre).  MODE 2: plain column sum with the
// GEMM A-prologue (dropout mask x row scale).
// ------------------------------------------------------------------------------------------------
struct ChanArgs {
    const float* a; const float* b;          // MODE0: a=y.  MODE1: a=dz, b=y.  MODE2: a=A
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    double* ws; float* out; float* part;      // ws [2C] doubles + part [grid][2C] floats (MODE0/1); out [C] floats (MODE2)
    float* masked;                            // MODE2: optional [M,C] copy of a x dropmask x rowscale (the masked upstream gradient)
    long lda; int M, C; int act;
    float drop_p; uint32_t k0, k1, thresh; float inv_keep; int rows_per_sample;
    const float* rowscale; int rows_per_scale;
    const uint32_t* seed;                     // optional device-side dropout seed (see gemm.hip)
    int groups;                               // BN over `groups` consecutive row groups of M rows each (MODE0/1: grid.y = group;
                                              // mean/rstd/ws/part are per group); apply kernels: M = all rows, rows_per_group below
    long rows_per_group;
    int affine_stride;                        // 0: one gamma/beta/running-stat row shared by all groups; C: one row per group
                                              // (domain-specific normalisation: group g uses parameter row g)
};

__device__ __forceinline__ float act_grad(int act, float pre) {
    return act == MDVIT_ACT_HSWISH ? hswish_grad_f(pre) : (act == MDVIT_ACT_RELU ? (pre > 0.f ? 1.f : 0.f) : 1.f);
}
__device__ __forceinline__ float act_fwd(int act, float pre) {
    return act == MDVIT_ACT_HSWISH ? hswish_f(pre) : (act == MDVIT_ACT_RELU ? fmaxf(pre, 0.f) : pre);
}

template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(ChanArgs p) {
    extern __shared__ float s_acc[];          // [2*C]
    const uint32_t k0e = p.k0 ^ (p.seed ? p.seed[0] : 0u), k1e = p.k1 + (p.seed ? p.seed[1] : 0u);
    const int QC = p.C >> 2;
    for (int i = threadIdx.x; i < 2 * p.C; i += blockDim.x) s_acc[i] = 0.f;
    __syncthreads();
    const long T = (long)gridDim.x * blockDim.x;
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(t0 % QC), c = q * 4;
    const long total = (long)p.M * QC;
    const int grp = MODE == 2 ? 0 : (int)blockIdx.y;        // BN group: rows [grp*M, (grp+1)*M)
    const long row0 = (long)grp * p.M;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4], rs[4], ga[4], be[4];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = p.mean[grp * p.C + c + j]; rs[j] = p.rstd[grp * p.C + c + j]; ga[j] = p.gamma[grp * p.affine_stride + c + j]; be[j] = p.beta[grp * p.affine_stride + c + j]; }
    }
    // rows t0 / QC, + T / QC, ... in increasing order (grid * 256 is a multiple of QC: chan_grid), FOUR rows' loads in flight per trip: one row per trip with a 64-bit
    // division in front of its load was a chain of dependent round trips (the 1 M-row reductions of the peer heads: 64 trips per thread, 75 us for 140 MB)
    (void)total;
    const long rstep = T / QC;
    auto one_row = [&](long row, const float4 av, const float4 yv) __attribute__((always_inline)) {
        const float a4[4] = {av.x, av.y, av.z, av.w};
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] += a4[j]; s2[j] += a4[j] * a4[j]; }
        } else if (MODE == 1) {
            const float y4[4] = {yv.x, yv.y, yv.z, yv.w};
            float ds[4] = {1.f, 1.f, 1.f, 1.f};
            if (p.drop_p > 0.f) {
                const float4 d4 = mdvit_drop_scale4(k0e, k1e, (uint32_t)((row / p.rows_per_sample) * p.C + c), p.thresh, p.inv_keep);
                ds[0] = d4.x; ds[1] = d4.y; ds[2] = d4.z; ds[3] = d4.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (y4[j] - mu[j]) * rs[j];
                const float g = a4[j] * act_grad(p.act, xh * ga[j] + be[j]) * ds[j];
                s1[j] += g; s2[j] += g * xh;
            }
        } else {
            const float rsc_ = p.rowscale ? p.rowscale[row / p.rows_per_scale] : 1.f;
            float v4[4];
            float ds[4] = {rsc_, rsc_, rsc_, rsc_};
            if (p.drop_p > 0.f) {
                const float4 d4 = mdvit_drop_scale4(k0e, k1e, (uint32_t)(row * p.C + c), p.thresh, p.inv_keep);
                ds[0] *= d4.x; ds[1] *= d4.y; ds[2] *= d4.z; ds[3] *= d4.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = a4[j] * ds[j];
                s1[j] += v; v4[j] = v;
            }
            if (p.masked) *reinterpret_cast<float4*>(p.masked + row * p.C + c) = make_float4(v4[0], v4[1], v4[2], v4[3]);
        }
    };
    long r = t0 / QC;
    for (; r + 3 * rstep < p.M; r += 4 * rstep) {
        float4 av[4], yv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long row = row0 + r + k * rstep;
            av[k] = *reinterpret_cast<const float4*>(p.a + row * p.lda + c);
            yv[k] = MODE == 1 ? *reinterpret_cast<const float4*>(p.b + row * p.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) one_row(row0 + r + k * rstep, av[k], yv[k]);
    }
    for (; r < p.M; r += rstep) {
        const long row = row0 + r;
        const float4 av = *reinterpret_cast<const float4*>(p.a + row * p.lda + c);
        const float4 yv = MODE == 1 ? *reinterpret_cast<const float4*>(p.b + row * p.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        one_row(row, av, yv);
    }
    if (MODE == 2) {
        if (!p.out) return;                   // masked copy only
        // fixed-order fold of the threads that share a channel quad (as MODE 0/1 below), one partial row per workgroup
        float* s_p4 = s_acc + 2 * p.C;        // [blockDim][4]
        *reinterpret_cast<float4*>(&s_p4[threadIdx.x * 4]) = make_float4(s1[0], s1[1], s1[2], s1[3]);
        __syncthreads();
        const int bq0 = (int)(((long)blockIdx.x * blockDim.x) % QC);
        for (int qq = threadIdx.x; qq < QC; qq += blockDim.x) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int tt = (qq - bq0 + QC) % QC; tt < (int)blockDim.x; tt += QC) {
                const float4 v = *reinterpret_cast<const float4*>(&s_p4[tt * 4]);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            *reinterpret_cast<float4*>(p.part + (long)blockIdx.x * p.C + qq * 4) = a;
        }
        return;
    }
    // MODE 0/1: bitwise-reproducible reduction.  Every thread parks its 8 partials in LDS; the partials of one
    // channel quad are then added in increasing thread order, and each block writes ONE partial row
    // part[block][2C]; chan_finalize_kernel adds the rows in block order (double).
    float* s_part = s_acc + 2 * p.C;          // [blockDim][8]
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_part[threadIdx.x * 8 + j] = s1[j]; s_part[threadIdx.x * 8 + 4 + j] = s2[j]; }
    __syncthreads();
    const int blk_q0 = (int)(((long)blockIdx.x * blockDim.x) % QC);      // quad owned by thread 0 of this block
    for (int qq = threadIdx.x; qq < QC; qq += blockDim.x) {
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        for (int tt = (qq - blk_q0 + QC) % QC; tt < (int)blockDim.x; tt += QC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { a1[j] += s_part[tt * 8 + j]; a2[j] += s_part[tt * 8 + 4 + j]; }
        }
        float* dst = p.part + ((long)grp * gridDim.x + blockIdx.x) * 2 * p.C;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dst[qq * 4 + j] = a1[j]; dst[p.C + qq * 4 + j] = a2[j]; }
    }
}

// ws[i] = sum over blocks of part[block][i], i < 2C, in double and in a FIXED order:
// 32 columns x 8 row lanes per workgroup; lane r adds rows r, r+8, ... then the 8 lane sums are added 0..7.
__global__ __launch_bounds__(256) void chan_finalize_kernel(const float* __restrict__ part, double* __restrict__ ws, int nblk, int C2) {
    __shared__ double s_sum[8][33];
    part += (long)blockIdx.y * nblk * C2;          // one BN group per grid.y
    ws += (long)blockIdx.y * C2;
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;
    double s = 0.0;
    if (i < C2) {
        int b = rl;
        for (; b + 24 < nblk; b += 32) {        // 4 independent loads in flight
            const float v0 = part[(long)b * C2 + i], v1 = part[(long)(b + 8) * C2 + i];
            const float v2 = part[(long)(b + 16) * C2 + i], v3 = part[(long)(b + 24) * C2 + i];
            s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
        }
        for (; b < nblk; b += 8) s += (double)part[(long)b * C2 + i];
    }
    s_sum[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && i < C2) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += s_sum[r][cl];
        ws[i] = t;
    }
}

// chan_finalize_kernel's column sums + the statistics' finalisation as ONE launch (BatchNorm forward statistics; one launch less per BatchNorm on the main stream): a workgroup owns
// 16 channels = the 32 columns {S1[c], S2[c]}; per group the partial rows are summed in double by 32 row lanes per column (lane r adds rows r, r + 32, ...,
// then the 32 lane sums are added 0..31: a FIXED order), then 16 threads turn the sums into mean / rstd / running statistics for their channel, group after group (the
// running statistics see the same update sequence as `groups` consecutive forwards).
__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const float* __restrict__ part, double* __restrict__ ws, int nblk, float* mean, float* rstd,
                                                                  float* rmean, float* rvar, int64_t* nbt, int M, int C, float eps, float momentum, int groups,
                                                                  int per_group) {
    __shared__ double s_sum[32][33];
    __shared__ double s_tot[32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 16;
    const int c = c0 + (cl & 15);
    const int col = (cl < 16 ? 0 : C) + c;             // column of the partial rows: S1 | S2
    const int C2 = 2 * C;
    float rm = 0.f, rv = 0.f;
    const bool owner = threadIdx.x < 16 && c < C;
    if (owner && !per_group) { rm = rmean ? rmean[c] : 0.f; rv = rvar ? rvar[c] : 0.f; }
    for (int g = 0; g < groups; ++g) {
        const float* pg = part + (long)g * nblk * C2;
        double s = 0.0;
        if (c < C) {
            int b = rl;
            for (; b + 96 < nblk; b += 128) {       // 4 independent loads in flight
                const float v0 = pg[(long)b * C2 + col], v1 = pg[(long)(b + 32) * C2 + col];
                const float v2 = pg[(long)(b + 64) * C2 + col], v3 = pg[(long)(b + 96) * C2 + col];
                s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
            }
            for (; b < nblk; b += 32) s += (double)pg[(long)b * C2 + col];
        }
        s_sum[rl][cl] = s;
        __syncthreads();
        if (rl == 0) {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 32; ++r) t += s_sum[r][cl];
            s_tot[cl] = t;
            if (c < C) ws[(long)g * C2 + col] = t;
        }
        __syncthreads();
        if (owner) {
            const double m = s_tot[cl] / M;
            double var = s_tot[cl + 16] / M - m * m;
            if (var < 0.0) var = 0.0;
            mean[g * C + c] = (float)m;
            rstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
            const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
            if (per_group) {
                if (rmean) {
                    rmean[g * C + c] = (1.f - momentum) * rmean[g * C + c] + momentum * (float)m;
                    rvar[g * C + c] = (1.f - momentum) * rvar[g * C + c] + momentum * (float)unb;
                }
            } else {
                rm = (1.f - momentum) * rm + momentum * (float)m;
                rv = (1.f - momentum) * rv + momentum * (float)unb;
            }
        }
    }
    if (owner && !per_group && rmean) { rmean[c] = rm; rvar[c] = rv; }
    if (blockIdx.x == 0 && nbt) {
        if (per_group) { if ((int)threadIdx.x < groups) nbt[threadIdx.x] += 1; }
        else if (threadIdx.x == 0) *nbt += groups;
    }
}

__global__ void bn_eval_prep_kernel(const float* rm, const float* rv, float* mean, float* rstd, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { mean[c] = rm[c]; rstd[c] = 1.0f / sqrtf(rv[c] + eps); }
}

// z = act((y-mean)*rstd*gamma+beta) * drop2d.  grid = (x, groups); the thread count along x is a multiple of C/4, so a
// thread keeps ONE channel quad of ONE statistics group: its 16 normalisation constants live in registers and the row
// loop is pure streaming (no per-element parameter loads, no 64-bit divisions).  p.M = rows per group.
__global__ __launch_bounds__(256) void bn_apply_kernel(ChanArgs p, float* __restrict__ z) {
    const uint32_t k0e = p.k0 ^ (p.seed ? p.seed[0] : 0u), k1e = p.k1 + (p.seed ? p.seed[1] : 0u);
    const int QC = p.C >> 2;
    const long T = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(t0 % QC) * 4, grp = blockIdx.y;
    const long row0 = (long)grp * p.M, rstep = T / QC;
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = p.mean[grp * p.C + c + j]; rs[j] = p.rstd[grp * p.C + c + j]; ga[j] = p.gamma[grp * p.affine_stride + c + j]; be[j] = p.beta[grp * p.affine_stride + c + j]; }
    for (long r = t0 / QC; r < p.M; r += rstep) {
        const long row = row0 + r;
        const float4 yv = *reinterpret_cast<const float4*>(p.a + row * p.C + c);
        const float y4[4] = {yv.x, yv.y, yv.z, yv.w};
        float o[4];
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop_p > 0.f) {
            const float4 d4 = mdvit_drop_scale4(k0e, k1e, (uint32_t)((row / p.rows_per_sample) * p.C + c), p.thresh, p.inv_keep);
            ds[0] = d4.x; ds[1] = d4.y; ds[2] = d4.z; ds[3] = d4.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = act_fwd(p.act, (y4[j] - mu[j]) * rs[j] * ga[j] + be[j]) * ds[j];
        *reinterpret_cast<float4*>(z + row * p.C + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// dy = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M)   (training)   |   gamma*rstd*g   (eval);  same thread layout as bn_apply
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(ChanArgs p, float* __restrict__ dy, float* dgamma, float* dbeta, int training) {
    const uint32_t k0e = p.k0 ^ (p.seed ? p.seed[0] : 0u), k1e = p.k1 + (p.seed ? p.seed[1] : 0u);
    const int QC = p.C >> 2;
    const long T = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(t0 % QC) * 4, grp = blockIdx.y;
    const long row0 = (long)grp * p.M, rstep = T / QC;
    const double invM = 1.0 / (double)p.M;
    const double* wsg = p.ws + (long)grp * 2 * p.C;
    float mu[4], rs[4], ga[4], be[4], sg[4], sgx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mu[j] = p.mean[grp * p.C + c + j]; rs[j] = p.rstd[grp * p.C + c + j]; ga[j] = p.gamma[grp * p.affine_stride + c + j]; be[j] = p.beta[grp * p.affine_stride + c + j];
        sg[j] = training ? (float)(wsg[c + j] * invM) : 0.f;
        sgx[j] = training ? (float)(wsg[p.C + c + j] * invM) : 0.f;
    }
    for (long r = t0 / QC; r < p.M; r += rstep) {
        const long row = row0 + r;
        const float4 dv = *reinterpret_cast<const float4*>(p.a + row * p.C + c);
        const float4 yv = *reinterpret_cast<const float4*>(p.b + row * p.C + c);
        const float d4[4] = {dv.x, dv.y, dv.z, dv.w}, y4[4] = {yv.x, yv.y, yv.z, yv.w};
        float o[4];
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop_p > 0.f) {
            const float4 q4 = mdvit_drop_scale4(k0e, k1e, (uint32_t)((row / p.rows_per_sample) * p.C + c), p.thresh, p.inv_keep);
            ds[0] = q4.x; ds[1] = q4.y; ds[2] = q4.z; ds[3] = q4.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (y4[j] - mu[j]) * rs[j];
            const float g = d4[j] * act_grad(p.act, xh * ga[j] + be[j]) * ds[j];
            o[j] = ga[j] * rs[j] * (g - sg[j] - xh * sgx[j]);
        }
        *reinterpret_cast<float4*>(dy + row * p.C + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (blockIdx.x == 0 && p.affine_stride) {            // per-group parameters: each group's sums are its own gradient row
        for (int cc = threadIdx.x; cc < p.C; cc += blockDim.x) {
            dbeta[grp * p.C + cc] = (float)wsg[cc]; dgamma[grp * p.C + cc] = (float)wsg[p.C + cc];
        }
    } else if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int cc = threadIdx.x; cc < p.C; cc += blockDim.x) {
            double sb = 0.0, sgm = 0.0;
            for (int g = 0; g < p.groups; ++g) { sb += p.ws[(long)g * 2 * p.C + cc]; sgm += p.ws[(long)g * 2 * p.C + p.C + cc]; }
            dbeta[cc] = (float)sb; dgamma[cc] = (float)sgm;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rowdot: y[m] = x[m,:].w + b    (wave per row)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int M, int K, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    for (int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < M; row += nw) {
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += x[(long)row * ldx + k] * w[k];
        s = wave_sum(s);
        if (lane == 0) {
            if (b) s += b[0];
            y[row] = accumulate ? y[row] + s : s;
        }
    }
}

template <int VPT>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                         const float* __restrict__ dy, float* __restrict__ dx, long lddx,
                                                         float* __restrict__ part, int has_db, int M, int K) {
    __shared__ float s_dw[4][64 * VPT];
    __shared__ float s_db[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    float adw[VPT], wv[VPT];
    float adb = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) { adw[j] = 0.f; const int k = lane + 64 * j; wv[j] = k < K ? w[k] : 0.f; }
    const bool want_w = part != nullptr;          // data-gradient-only call: x is not even read
    for (int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < M; row += nw) {
        const float g = dy[row];
        adb += g;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int k = lane + 64 * j;
            if (k < K) {
                if (want_w) adw[j] += g * x[(long)row * ldx + k];
                if (dx) dx[(long)row * lddx + k] = g * wv[j];
            }
        }
    }
    if (!want_w) return;
#pragma unroll
    for (int j = 0; j < VPT; ++j) s_dw[wave][lane + 64 * j] = adw[j];
    if (lane == 0) s_db[wave] = adb;     // every lane of a wave holds the same adb
    __syncthreads();
    float* row = part + (long)blockIdx.x * (K + 1);          // [dw (K) | db (1)]
    for (int k = threadIdx.x; k < K; k += blockDim.x) row[k] = s_dw[0][k] + s_dw[1][k] + s_dw[2][k] + s_dw[3][k];
    if (threadIdx.x == 0) row[K] = has_db ? s_db[0] + s_db[1] + s_db[2] + s_db[3] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm -> activation -> Dropout2d -> 1-output 1x1 conv as ONE op (the tail of the peer heads, Decoders.py:304-311,333-336 and
// Utils/_deeplab.py head: the [B h w, 512] tensor between the norm and the 512 -> 1 conv never exists).
//   fwd :  low[m] = b + sum_c w[c] * act((y[m,c] - mean_c) rstd_c gamma_c + beta_c) * drop2d(sample(m), c)        (a wave per row)
//   bwd :  with G[m,c] = g[m] w[c] act'(pre) drop:  S1_c = sum_m G,  S2_c = sum_m G xhat,  S3_c = sum_m g[m] z[m,c] (= dw),  S4 = sum_m g (= db)
//          dy[m,c] = gamma_c rstd_c (G - S1_c / M - xhat S2_c / M)   (training; eval: gamma rstd G),  dgamma = S2, dbeta = S1
// Against bn_apply + rowdot / rowdot_bwd + bn_bwd the forward reads y once instead of moving 4 tensors of its size, the backward moves 3 instead of 7.
// ------------------------------------------------------------------------------------------------
template <int VPL>          // VPL float4 per lane: C = 256 * VPL
__global__ __launch_bounds__(256) void bn_rowdot_fwd_kernel(ChanArgs p, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ low) {
    const uint32_t k0e = p.k0 ^ (p.seed ? p.seed[0] : 0u), k1e = p.k1 + (p.seed ? p.seed[1] : 0u);
    const int lane = threadIdx.x & 63;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    float sc[VPL][4], sh[VPL][4], wv[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (lane + 64 * v) * 4 + j;
            const float rs = p.rstd[c] * p.gamma[c];
            sc[v][j] = rs; sh[v][j] = p.beta[c] - p.mean[c] * rs; wv[v][j] = w[c];
        }
    const float b0 = bias ? bias[0] : 0.f;
    for (long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < p.M; row += nw) {
        float4 yv[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) yv[v] = *reinterpret_cast<const float4*>(p.a + row * p.C + (lane + 64 * v) * 4);
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const float y4[4] = {yv[v].x, yv[v].y, yv[v].z, yv[v].w};
            float ds[4] = {1.f, 1.f, 1.f, 1.f};
            if (p.drop_p > 0.f) {
                const float4 d4 = mdvit_drop_scale4(k0e, k1e, (uint32_t)((row / p.rows_per_sample) * p.C + (lane + 64 * v) * 4), p.thresh, p.inv_keep);
                ds[0] = d4.x; ds[1] = d4.y; ds[2] = d4.z; ds[3] = d4.w;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) s += wv[v][j] * (act_fwd(p.act, fmaf(y4[j], sc[v][j], sh[v][j])) * ds[j]);
        }
        s = wave_sum(s);
        if (lane == 0) low[row] = s + b0;
    }
}

// per-channel sums of the backward (thread = one channel quad, fixed-order fold as chan_reduce_kernel): partial row [S1 (C) | S2 (C) | S3 (C) | S4, 0, 0, 0]
__global__ __launch_bounds__(256) void bn_rowdot_reduce_kernel(ChanArgs p, const float* __restrict__ g, const float* __restrict__ w, int want_w) {
    extern __shared__ float s_part[];         // [blockDim][13]
    const uint32_t k0e = p.k0 ^ (p.seed ? p.seed[0] : 0u), k1e = p.k1 + (p.seed ? p.seed[1] : 0u);
    const int QC = p.C >> 2;
    const long T = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(t0 % QC), c = q * 4;
    const long total = (long)p.M * QC;
    float mu[4], rs[4], ga[4], be[4], wv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = p.mean[c + j]; rs[j] = p.rstd[c + j]; ga[j] = p.gamma[c + j]; be[j] = p.beta[c + j]; wv[j] = w[c + j]; }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, s3[4] = {0.f, 0.f, 0.f, 0.f}, s4 = 0.f;
    (void)total;
    const long rstep = T / QC;                // (grid * 256 is a multiple of QC: chan_grid) -- rows in increasing order, four rows' loads in flight, as chan_reduce_kernel
    auto one_row = [&](long row, const float gr, const float4 yv) __attribute__((always_inline)) {
        const float y4[4] = {yv.x, yv.y, yv.z, yv.w};
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop_p > 0.f) {
            const float4 d4 = mdvit_drop_scale4(k0e, k1e, (uint32_t)((row / p.rows_per_sample) * p.C + c), p.thresh, p.inv_keep);
            ds[0] = d4.x; ds[1] = d4.y; ds[2] = d4.z; ds[3] = d4.w;
        }
        if (q == 0) s4 += gr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (y4[j] - mu[j]) * rs[j], pre = xh * ga[j] + be[j];
            const float G = gr * wv[j] * act_grad(p.act, pre) * ds[j];
            s1[j] += G; s2[j] += G * xh;
            if (want_w) s3[j] += gr * (act_fwd(p.act, pre) * ds[j]);
        }
    };
    long r = t0 / QC;
    for (; r + 3 * rstep < p.M; r += 4 * rstep) {
        float gr[4]; float4 yv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { gr[k] = g[r + k * rstep]; yv[k] = *reinterpret_cast<const float4*>(p.a + (r + k * rstep) * p.C + c); }
#pragma unroll
        for (int k = 0; k < 4; ++k) one_row(r + k * rstep, gr[k], yv[k]);
    }
    for (; r < p.M; r += rstep) one_row(r, g[r], *reinterpret_cast<const float4*>(p.a + r * p.C + c));
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_part[threadIdx.x * 13 + j] = s1[j]; s_part[threadIdx.x * 13 + 4 + j] = s2[j]; s_part[threadIdx.x * 13 + 8 + j] = s3[j]; }
    s_part[threadIdx.x * 13 + 12] = s4;
    __syncthreads();
    const int blk_q0 = (int)(((long)blockIdx.x * blockDim.x) % QC);
    float* dst = p.part + (long)blockIdx.x * (3 * p.C + 4);
    for (int qq = threadIdx.x; qq < QC; qq += blockDim.x) {
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f}, a3[4] = {0.f, 0.f, 0.f, 0.f}, a4 = 0.f;
        for (int tt = (qq - blk_q0 + QC) % QC; tt < (int)blockDim.x; tt += QC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { a1[j] += s_part[tt * 13 + j]; a2[j] += s_part[tt * 13 + 4 + j]; a3[j] += s_part[tt * 13 + 8 + j]; }
            a4 += s_part[tt * 13 + 12];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { dst[qq * 4 + j] = a1[j]; dst[p.C + qq * 4 + j] = a2[j]; dst[2 * p.C + qq * 4 + j] = a3[j]; }
        if (qq == 0) { dst[3 * p.C] = a4; dst[3 * p.C + 1] = 0.f; dst[3 * p.C + 2] = 0.f; dst[3 * p.C + 3] = 0.f; }
    }
}

__global__ __launch_bounds__(256) void bn_rowdot_bwd_apply_kernel(ChanArgs p, const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dy,
                                                                 float* dgamma, float* dbeta, float* dw, float* db, int training) {
    const uint32_t k0e = p.k0 ^ (p.seed ? p.seed[0] : 0u), k1e = p.k1 + (p.seed ? p.seed[1] : 0u);
    const int QC = p.C >> 2;
    const long T = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(t0 % QC) * 4;
    const long rstep = T / QC;
    const double invM = 1.0 / (double)p.M;
    float mu[4], rs[4], ga[4], be[4], wv[4], sg[4], sgx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mu[j] = p.mean[c + j]; rs[j] = p.rstd[c + j]; ga[j] = p.gamma[c + j]; be[j] = p.beta[c + j]; wv[j] = w[c + j];
        sg[j] = training ? (float)(p.ws[c + j] * invM) : 0.f;
        sgx[j] = training ? (float)(p.ws[p.C + c + j] * invM) : 0.f;
    }
    for (long row = t0 / QC; row < p.M; row += rstep) {
        const float gr = g[row];
        const float4 yv = *reinterpret_cast<const float4*>(p.a + row * p.C + c);
        const float y4[4] = {yv.x, yv.y, yv.z, yv.w};
        float ds[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop_p > 0.f) {
            const float4 q4 = mdvit_drop_scale4(k0e, k1e, (uint32_t)((row / p.rows_per_sample) * p.C + c), p.thresh, p.inv_keep);
            ds[0] = q4.x; ds[1] = q4.y; ds[2] = q4.z; ds[3] = q4.w;
        }
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (y4[j] - mu[j]) * rs[j];
            const float G = gr * wv[j] * act_grad(p.act, xh * ga[j] + be[j]) * ds[j];
            o[j] = ga[j] * rs[j] * (G - sg[j] - xh * sgx[j]);
        }
        *reinterpret_cast<float4*>(dy + row * p.C + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (blockIdx.x == 0 && dgamma) {
        for (int cc = threadIdx.x; cc < p.C; cc += blockDim.x) { dbeta[cc] = (float)p.ws[cc]; dgamma[cc] = (float)p.ws[p.C + cc]; dw[cc] = (float)p.ws[2 * p.C + cc]; }
        if (threadIdx.x == 0 && db) db[0] = (float)p.ws[3 * p.C];
    }
}

int chan_grid(long M, int C, int max_blocks) {
    // grid*256 must be a multiple of C/4
    const int QC = C / 4;
    int g = 1;
    {
        int a = QC, b = 256;
        while (b) { int t = a % b; a = b; b = t; }
        g = QC / a;                       // smallest grid with (grid*256) % QC == 0
    }
    long want = ((long)M * QC + 256L * 8 - 1) / (256L * 8);
    if (want > max_blocks) want = max_blocks;
    if (want < 1) want = 1;
    long grid = (want + g - 1) / g * g;
    return (int)grid;
}

void fill_drop(ChanArgs& a, float p, uint32_t k0, uint32_t k1, int rows_per_sample) {
    a.drop_p = p; a.k0 = k0; a.k1 = k1;
    a.thresh = mdvit_drop_thresh(p);
    a.inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
    a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
}

}  // namespace

#define LN_DISPATCH(KERN, C, ...)                                                                    \
    do {                                                                                             \
        if (C <= 64) hipLaunchKernelGGL((KERN<1>), grid, dim3(256), 0, s, __VA_ARGS__);              \
        else if (C <= 128) hipLaunchKernelGGL((KERN<2>), grid, dim3(256), 0, s, __VA_ARGS__);        \
        else if (C <= 256) hipLaunchKernelGGL((KERN<4>), grid, dim3(256), 0, s, __VA_ARGS__);        \
        else if (C <= 512) hipLaunchKernelGGL((KERN<8>), grid, dim3(256), 0, s, __VA_ARGS__);        \
        else hipLaunchKernelGGL((KERN<16>), grid, dim3(256), 0, s, __VA_ARGS__);                     \
    } while (0)

extern "C" int mdvit_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                   int32_t M, int32_t C, int32_t groups, float eps, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(M > 0 && C > 0 && C <= 1024, MDVIT_E_SHAPE, "layernorm_fwd: need 0 < C <= 1024 (M=%d C=%d)", M, C);
    MDVIT_CHECK_ARG(groups > 0 && M % groups == 0, MDVIT_E_SHAPE, "layernorm_fwd: M=%d is not a multiple of groups=%d", M, groups);
    const int Mg = M / groups;                 // rows per parameter group (kernels take the per-group row count; grid.y = group)
    if (C == 64 || C == 128 || C == 320 || C == 512) {
        dim3 grid16(min(cdiv(Mg, 16), max(1, 4096 / groups)), groups);
        if (C == 64) hipLaunchKernelGGL((ln_fwd16_kernel<1>), grid16, dim3(256), 0, s, x, gamma, beta, y, mean, rstd, Mg, eps);
        else if (C == 128) hipLaunchKernelGGL((ln_fwd16_kernel<2>), grid16, dim3(256), 0, s, x, gamma, beta, y, mean, rstd, Mg, eps);
        else if (C == 320) hipLaunchKernelGGL((ln_fwd16_kernel<5>), grid16, dim3(256), 0, s, x, gamma, beta, y, mean, rstd, Mg, eps);
        else hipLaunchKernelGGL((ln_fwd16_kernel<8>), grid16, dim3(256), 0, s, x, gamma, beta, y, mean, rstd, Mg, eps);
        MDVIT_LAUNCH_CHECK();
        return MDVIT_OK;
    }
    dim3 grid(min(cdiv(Mg, 4), max(1, 4096 / groups)), groups);
    LN_DISPATCH(ln_fwd_kernel, C, x, gamma, beta, y, mean, rstd, Mg, C, eps);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

// A second upstream-gradient partial for the NEXT LayerNorm backward call of this thread (block.hip: the C = 64 MLP backward in one kernel leaves dx as one partial
// per 256-wide hidden role; the LayerNorm in front of the MLP adds them while it reads them -- no separate sum pass).  EVERY public LayerNorm backward entry takes
// (reads and clears) it as its first statement, before any argument check can return: a call that fails early must not leave the pointer armed for a later,
// unrelated LayerNorm backward of this thread.
static thread_local const float* g_ln_dy2 = nullptr;
void mdvit_layernorm_bwd_next_dy2(const float* dy2) { g_ln_dy2 = dy2; }
static inline const float* ln_take_dy2() { const float* p = g_ln_dy2; g_ln_dy2 = nullptr; return p; }

static int layernorm_bwd_impl(const float* dy, const float* dy2, const float* x, const float* gamma, const float* mean, const float* rstd,
                              const float* add, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                              int32_t M, int32_t C, int32_t groups, const LnMasked& mk, void* stream, int* defer_nblk = nullptr);

// dgamma / dbeta: [groups, C] each
extern "C" int mdvit_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                   const float* add, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                   int32_t M, int32_t C, int32_t groups, void* stream) {
    const float* dy2 = ln_take_dy2();
    LnMasked mk; memset(&mk, 0, sizeof(mk));
    return layernorm_bwd_impl(dy, dy2, x, gamma, mean, rstd, add, dx, dgamma, dbeta, ws, ws_bytes, M, C, groups, mk, stream);
}

extern "C" int mdvit_layernorm_bwd_masked(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                          const float* add, float* dx, float* dx_masked, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                          int32_t M, int32_t C, int32_t groups, float drop_p, uint32_t key0, uint32_t key1, const float* rowscale,
                                          int32_t rows_per_scale, const uint32_t* seed, void* stream) {
    const float* dy2 = ln_take_dy2();
    MDVIT_CHECK_ARG(C == 64 || C == 128 || C == 320 || C == 512, MDVIT_E_SHAPE, "layernorm_bwd_masked: C=%d not built (64/128/320/512)", C);
    MDVIT_CHECK_ARG(dx_masked != nullptr && drop_p >= 0.f && drop_p < 1.f && (rowscale == nullptr || rows_per_scale > 0), MDVIT_E_SHAPE,
                    "layernorm_bwd_masked: bad mask arguments");
    MDVIT_CHECK_ARG((long)M * C < (1L << 32), MDVIT_E_SHAPE, "layernorm_bwd_masked: dropout index space exceeds 2^32");
    LnMasked mk; memset(&mk, 0, sizeof(mk));
    mk.out = dx_masked; mk.drop_p = drop_p; mk.k0 = key0; mk.k1 = key1; mk.thresh = mdvit_drop_thresh(drop_p);
    mk.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; mk.rowscale = rowscale; mk.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1; mk.seed = seed;
    return layernorm_bwd_impl(dy, dy2, x, gamma, mean, rstd, add, dx, dgamma, dbeta, ws, ws_bytes, M, C, groups, mk, stream);
}

// The LayerNorm backward WITHOUT its second stage: the partial rows [groups][*nblk][dgamma | dbeta] stay in `ws`; the caller adds them up
// (mdvit_reduce_partials_batched2_acc), on any stream ordered after this one -- block.hip does it on the weight-gradient stream.
int mdvit_layernorm_bwd_parts(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* add, float* dx,
                              float* dx_masked, void* ws, size_t ws_bytes, int M, int C, int groups, float drop_p, uint32_t key0, uint32_t key1,
                              const float* rowscale, int rows_per_scale, const uint32_t* seed, hipStream_t stream, int* nblk) {
    const float* dy2 = ln_take_dy2();
    MDVIT_CHECK_ARG(C == 64 || C == 128 || C == 320 || C == 512, MDVIT_E_SHAPE, "layernorm_bwd_parts: C=%d not built (64/128/320/512)", C);
    MDVIT_CHECK_ARG(nblk != nullptr && ws != nullptr, MDVIT_E_SHAPE, "layernorm_bwd_parts: null argument");
    LnMasked mk; memset(&mk, 0, sizeof(mk));
    if (dx_masked) {
        MDVIT_CHECK_ARG((long)M * C < (1L << 32), MDVIT_E_SHAPE, "layernorm_bwd_parts: dropout index space exceeds 2^32");
        mk.out = dx_masked; mk.drop_p = drop_p; mk.k0 = key0; mk.k1 = key1; mk.thresh = mdvit_drop_thresh(drop_p);
        mk.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; mk.rowscale = rowscale; mk.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1; mk.seed = seed;
    }
    float dummy;
    return layernorm_bwd_impl(dy, dy2, x, gamma, mean, rstd, add, dx, &dummy, &dummy, ws, ws_bytes, M, C, groups, mk, stream, nblk);
}

static int layernorm_bwd_impl(const float* dy, const float* dy2, const float* x, const float* gamma, const float* mean, const float* rstd,
                              const float* add, float* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                              int32_t M, int32_t C, int32_t groups, const LnMasked& mk_in, void* stream, int* defer_nblk) {
    hipStream_t s = (hipStream_t)stream;
    LnMasked mk = mk_in;
    mk.dy2 = dy2;
    MDVIT_CHECK_ARG(mk.dy2 == nullptr || ((C == 64 || C == 128 || C == 320 || C == 512) && aligned16(mk.dy2)), MDVIT_E_SHAPE, "layernorm_bwd: a second dy partial needs C in 64/128/320/512");
    MDVIT_CHECK_ARG(M > 0 && C > 0 && C <= 1024, MDVIT_E_SHAPE, "layernorm_bwd: need 0 < C <= 1024 (M=%d C=%d)", M, C);
    MDVIT_CHECK_ARG(groups > 0 && groups <= 64 && M % groups == 0, MDVIT_E_SHAPE, "layernorm_bwd: M=%d is not a multiple of groups=%d", M, groups);
    const int Mg = M / groups;
    const bool want_params = dgamma != nullptr || dbeta != nullptr;        // both NULL: data gradient only (no partial sums, no reduction launch)
    float* part = want_params ? (float*)ws : nullptr;                      // [group][workgroup][dgamma | dbeta]
    int nblk;
    if (C == 64 || C == 128 || C == 320 || C == 512) {
        // rows per workgroup: 64 (four passes of 16) where the grid fills the chip anyway; 16 for the short tensors of stages 2 / 3 (C >= 320: 4096-16384 rows at
        // bs=4), which otherwise run as 256 four-wave workgroups with four dependent load rounds each (MDVIT_LN_BWD_ROWS=64: the old grid, A/B)
        static const int rows_env = [] { const char* e = getenv("MDVIT_LN_BWD_ROWS"); return e ? atoi(e) : 0; }();
        const int rows_wg = rows_env > 0 ? rows_env : ((C >= 320 && (long)M * C <= (16384L * 512)) ? 16 : 64);
        nblk = min(cdiv(Mg, rows_wg), 1024 / groups);
        if (want_params) MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nblk * groups, 2 * C, "layernorm_bwd");
        dim3 grid16(nblk, groups);
        if (C == 64) hipLaunchKernelGGL((ln_bwd16_kernel<1>), grid16, dim3(256), 0, s, dy, x, gamma, mean, rstd, add, dx, part, Mg, mk);
        else if (C == 128) hipLaunchKernelGGL((ln_bwd16_kernel<2>), grid16, dim3(256), 0, s, dy, x, gamma, mean, rstd, add, dx, part, Mg, mk);
        else if (C == 320) hipLaunchKernelGGL((ln_bwd16_kernel<5>), grid16, dim3(256), 0, s, dy, x, gamma, mean, rstd, add, dx, part, Mg, mk);
        else hipLaunchKernelGGL((ln_bwd16_kernel<8>), grid16, dim3(256), 0, s, dy, x, gamma, mean, rstd, add, dx, part, Mg, mk);
    } else {
        nblk = min(cdiv(Mg, 16), 1024 / groups);
        MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nblk * groups, 2 * C, "layernorm_bwd");
        part = (float*)ws;                                                 // (the generic-C kernel always writes its partial rows)
        dim3 grid(nblk, groups);
        LN_DISPATCH(ln_bwd_kernel, C, dy, x, gamma, mean, rstd, add, dx, part, Mg, C);
    }
    MDVIT_LAUNCH_CHECK();
    if (!want_params) return MDVIT_OK;
    if (defer_nblk) { *defer_nblk = nblk; return MDVIT_OK; }                                  // the caller runs the second stage
    return mdvit_reduce_partials_batched2(part, groups, nblk, C, dgamma, C, dbeta, s);      // fixed-order sums of the per-workgroup rows
}

constexpr int CHAN_MAX_BLOCKS = 512;

// M = rows of ONE group
static size_t bn_ws_bytes(int M, int C, int groups) {
    return (size_t)groups * (sizeof(double) * 2 * (size_t)C + sizeof(float) * 2 * (size_t)C * (size_t)chan_grid(M, C, CHAN_MAX_BLOCKS));
}

extern "C" size_t mdvit_bn_ws_bytes(int32_t M, int32_t C, int32_t groups) {
    if (M <= 0 || C <= 0 || C % 4 || groups <= 0 || M % groups) return 0;
    return bn_ws_bytes(M / groups, C, groups);
}

extern "C" int mdvit_bn_stats(const float* y, void* ws, size_t ws_bytes, float* mean, float* rstd, float* running_mean, float* running_var,
                              int64_t* nbt, int32_t M, int32_t C, int32_t groups, int32_t per_group_affine, float eps, float momentum, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && C <= 4096, MDVIT_E_SHAPE, "bn_stats: need C %% 4 == 0, C <= 4096 (M=%d C=%d)", M, C);
    MDVIT_CHECK_ARG(groups > 0 && M % groups == 0, MDVIT_E_SHAPE, "bn_stats: M=%d is not a multiple of groups=%d", M, groups);
    const int Mg = M / groups;
    MDVIT_CHECK_ARG(ws_bytes >= bn_ws_bytes(Mg, C, groups), MDVIT_E_WORKSPACE, "bn_stats: workspace too small");
    ChanArgs a; memset(&a, 0, sizeof(a));
    a.a = y; a.lda = C; a.M = Mg; a.C = C; a.groups = groups; a.rows_per_group = Mg;
    a.ws = (double*)ws; a.part = (float*)((double*)ws + 2 * (size_t)C * groups);
    const int grid = chan_grid(Mg, C, CHAN_MAX_BLOCKS);
    hipLaunchKernelGGL((chan_reduce_kernel<0>), dim3(grid, groups), dim3(256), sizeof(float) * (2 * C + 256 * 8), s, a);
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, s, a.part, a.ws, grid, mean, rstd, running_mean, running_var, nbt, Mg, C, eps, momentum,
                       groups, per_group_affine);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_bn_eval_prep(const float* rm, const float* rv, float* mean, float* rstd, int32_t C, float eps, void* stream) {
    hipLaunchKernelGGL(bn_eval_prep_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, rm, rv, mean, rstd, C, eps);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_bn_apply(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta, float* z,
                              int32_t M, int32_t C, int32_t groups, int32_t per_group_affine, int32_t act, float drop2d_p, uint32_t key0, uint32_t key1,
                              const uint32_t* seed, int32_t rows_per_sample, void* stream) {
    MDVIT_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, MDVIT_E_SHAPE, "bn_apply: need C %% 4 == 0 (M=%d C=%d)", M, C);
    MDVIT_CHECK_ARG(groups > 0 && M % groups == 0, MDVIT_E_SHAPE, "bn_apply: M=%d is not a multiple of groups=%d", M, groups);
    ChanArgs a; memset(&a, 0, sizeof(a));
    a.a = y; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.M = M; a.C = C; a.act = act;
    a.groups = groups; a.rows_per_group = M / groups; a.affine_stride = per_group_affine ? C : 0;
    a.M = M / groups;                          // rows per statistics group (grid.y = group)
    fill_drop(a, drop2d_p, key0, key1, rows_per_sample);
    a.seed = seed;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(chan_grid(a.M, C, max(1, 4096 / groups)), groups), dim3(256), 0, (hipStream_t)stream, a, z);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_bn_bwd(const float* dz, const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            float* dy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, int32_t M, int32_t C, int32_t groups,
                            int32_t per_group_affine, int32_t act, int32_t training, float drop2d_p, uint32_t key0, uint32_t key1, const uint32_t* seed, int32_t rows_per_sample,
                            void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && C <= 4096, MDVIT_E_SHAPE, "bn_bwd: need C %% 4 == 0, C <= 4096 (M=%d C=%d)", M, C);
    MDVIT_CHECK_ARG(groups > 0 && M % groups == 0, MDVIT_E_SHAPE, "bn_bwd: M=%d is not a multiple of groups=%d", M, groups);
    const int Mg = M / groups;
    MDVIT_CHECK_ARG(ws_bytes >= bn_ws_bytes(Mg, C, groups), MDVIT_E_WORKSPACE, "bn_bwd: workspace too small");
    ChanArgs a; memset(&a, 0, sizeof(a));
    a.a = dz; a.b = y; a.lda = C; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.M = Mg; a.C = C; a.act = act;
    a.groups = groups; a.rows_per_group = Mg; a.affine_stride = per_group_affine ? C : 0;
    a.ws = (double*)ws; a.part = (float*)((double*)ws + 2 * (size_t)C * groups);
    fill_drop(a, drop2d_p, key0, key1, rows_per_sample);
    a.seed = seed;
    const int grid = chan_grid(Mg, C, CHAN_MAX_BLOCKS);
    hipLaunchKernelGGL((chan_reduce_kernel<1>), dim3(grid, groups), dim3(256), sizeof(float) * (2 * C + 256 * 8), s, a);
    hipLaunchKernelGGL(chan_finalize_kernel, dim3(cdiv(2 * C, 32), groups), dim3(256), 0, s, a.part, a.ws, grid, 2 * C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(chan_grid(Mg, C, max(1, 4096 / groups)), groups), dim3(256), 0, s, a, dy, dgamma, dbeta, training);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" size_t mdvit_bn_rowdot_ws_bytes(int32_t M, int32_t C) {
    if (M <= 0 || C <= 0 || C % 4) return 0;
    return sizeof(double) * (3 * (size_t)C + 4) + sizeof(float) * (3 * (size_t)C + 4) * (size_t)chan_grid(M, C, CHAN_MAX_BLOCKS);
}

extern "C" int mdvit_bn_rowdot_fwd(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w, const float* b,
                                   float* low, int32_t M, int32_t C, int32_t act, float drop2d_p, uint32_t key0, uint32_t key1, const uint32_t* seed,
                                   int32_t rows_per_sample, void* stream) {
    MDVIT_CHECK_ARG(M > 0 && (C == 256 || C == 512 || C == 1024), MDVIT_E_SHAPE, "bn_rowdot_fwd: built for C = 256 / 512 / 1024 (M=%d C=%d)", M, C);
    MDVIT_CHECK_ARG(y && mean && rstd && gamma && beta && w && low && aligned16(y), MDVIT_E_SHAPE, "bn_rowdot_fwd: null or misaligned operand");
    ChanArgs a; memset(&a, 0, sizeof(a));
    a.a = y; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.M = M; a.C = C; a.act = act;
    fill_drop(a, drop2d_p, key0, key1, rows_per_sample);
    a.seed = seed;
    const dim3 grid(min(cdiv(M, 4), 8192));
    if (C == 256) hipLaunchKernelGGL((bn_rowdot_fwd_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, a, w, b, low);
    else if (C == 512) hipLaunchKernelGGL((bn_rowdot_fwd_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, a, w, b, low);
    else hipLaunchKernelGGL((bn_rowdot_fwd_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, a, w, b, low);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_bn_rowdot_bwd(const float* g, const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w,
                                   float* dy, float* dgamma, float* dbeta, float* dw, float* db, void* ws, size_t ws_bytes, int32_t M, int32_t C, int32_t act,
                                   int32_t training, float drop2d_p, uint32_t key0, uint32_t key1, const uint32_t* seed, int32_t rows_per_sample, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0 && C <= 4096, MDVIT_E_SHAPE, "bn_rowdot_bwd: need C %% 4 == 0, C <= 4096 (M=%d C=%d)", M, C);
    MDVIT_CHECK_ARG(g && y && mean && rstd && gamma && beta && w && dy && aligned16(y) && aligned16(dy), MDVIT_E_SHAPE, "bn_rowdot_bwd: null or misaligned operand");
    MDVIT_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr) && (dgamma == nullptr) == (dw == nullptr), MDVIT_E_SHAPE,
                    "bn_rowdot_bwd: dgamma, dbeta and dw must be all given or all NULL (data gradient only)");
    MDVIT_CHECK_ARG(ws && ws_bytes >= mdvit_bn_rowdot_ws_bytes(M, C), MDVIT_E_WORKSPACE, "bn_rowdot_bwd: workspace too small (mdvit_bn_rowdot_ws_bytes)");
    ChanArgs a; memset(&a, 0, sizeof(a));
    a.a = y; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.M = M; a.C = C; a.act = act;
    a.ws = (double*)ws; a.part = (float*)((double*)ws + 3 * (size_t)C + 4);
    fill_drop(a, drop2d_p, key0, key1, rows_per_sample);
    a.seed = seed;
    const int C2 = 3 * C + 4;
    if (training || dgamma) {
        const int grid = chan_grid(M, C, CHAN_MAX_BLOCKS);
        hipLaunchKernelGGL(bn_rowdot_reduce_kernel, dim3(grid), dim3(256), sizeof(float) * 256 * 13, s, a, g, w, dgamma != nullptr);
        hipLaunchKernelGGL(chan_finalize_kernel, dim3(cdiv(C2, 32), 1), dim3(256), 0, s, a.part, a.ws, grid, C2);
    }
    hipLaunchKernelGGL(bn_rowdot_bwd_apply_kernel, dim3(chan_grid(M, C, 4096)), dim3(256), 0, s, a, g, w, dy, dgamma, dbeta, dw, db, training);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

static int colsum_impl(const float* A, int64_t lda, float* out, float* masked, void* ws, size_t ws_bytes, int32_t M, int32_t N, float drop_p,
                       uint32_t key0, uint32_t key1, const float* rowscale, int32_t rows_per_scale, int32_t accumulate, const uint32_t* seed, void* stream, int* defer_nblk);

// column sums WITHOUT the second stage: the partial rows [*nblk][N] stay in `ws` (see mdvit_layernorm_bwd_parts)
int mdvit_colsum_parts(const float* A, long lda, float* masked, void* ws, size_t ws_bytes, int M, int N, float drop_p, uint32_t key0, uint32_t key1,
                       const float* rowscale, int rows_per_scale, const uint32_t* seed, hipStream_t stream, int* nblk) {
    float dummy;
    return colsum_impl(A, lda, &dummy, masked, ws, ws_bytes, M, N, drop_p, key0, key1, rowscale, rows_per_scale, 0, seed, stream, nblk);
}

extern "C" int mdvit_colsum_f32(const float* A, int64_t lda, float* out, float* masked, void* ws, size_t ws_bytes, int32_t M, int32_t N, float drop_p,
                                uint32_t key0, uint32_t key1, const float* rowscale, int32_t rows_per_scale, int32_t accumulate,
                                const uint32_t* seed, void* stream) {
    return colsum_impl(A, lda, out, masked, ws, ws_bytes, M, N, drop_p, key0, key1, rowscale, rows_per_scale, accumulate, seed, stream, nullptr);
}

static int colsum_impl(const float* A, int64_t lda, float* out, float* masked, void* ws, size_t ws_bytes, int32_t M, int32_t N, float drop_p,
                       uint32_t key0, uint32_t key1, const float* rowscale, int32_t rows_per_scale, int32_t accumulate, const uint32_t* seed, void* stream,
                       int* defer_nblk) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(M > 0 && N > 0 && N % 4 == 0 && N <= 8192 && lda % 4 == 0, MDVIT_E_SHAPE, "colsum: need N %% 4 == 0 (M=%d N=%d)", M, N);
    MDVIT_CHECK_ARG(out || masked, MDVIT_E_SHAPE, "colsum: nothing to produce (out and masked are both NULL)");
    MDVIT_CHECK_ARG((long)M * N < (1L << 32), MDVIT_E_SHAPE, "colsum: dropout index space exceeds 2^32");
    const int nblk = chan_grid(M, N, out ? 1024 : 4096);
    if (out) MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nblk, N, "colsum");
    ChanArgs a; memset(&a, 0, sizeof(a));
    a.a = A; a.lda = lda; a.M = M; a.C = N; a.out = out; a.masked = masked; a.part = (float*)ws;
    fill_drop(a, drop_p, key0, key1, 1);
    a.seed = seed;
    a.rowscale = rowscale; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    hipLaunchKernelGGL((chan_reduce_kernel<2>), dim3(nblk), dim3(256), sizeof(float) * (2 * N + 256 * 4), s, a);
    MDVIT_LAUNCH_CHECK();
    if (defer_nblk) { *defer_nblk = nblk; return MDVIT_OK; }
    return out ? mdvit_reduce_partials((const float*)ws, nblk, (long)N, N, out, 0, nullptr, accumulate, s) : MDVIT_OK;
}

extern "C" int mdvit_rowdot_fwd(const float* x, int64_t ldx, const float* w, const float* b, float* y, int32_t M, int32_t K,
                                int32_t accumulate, void* stream) {
    MDVIT_CHECK_ARG(M > 0 && K > 0, MDVIT_E_SHAPE, "rowdot_fwd: bad shape M=%d K=%d", M, K);
    hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(min(cdiv(M, 4), 8192)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, w, b, y, M, K, accumulate);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_rowdot_bwd(const float* x, int64_t ldx, const float* w, const float* dy, float* dx, int64_t lddx, float* dw, float* db,
                                void* ws, size_t ws_bytes, int32_t M, int32_t K, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(M > 0 && K > 0 && K <= 1024, MDVIT_E_SHAPE, "rowdot_bwd: need K <= 1024 (M=%d K=%d)", M, K);
    const int nblk = min(cdiv(M, 4), 1024);      // one row per wavefront per pass: small M (weight composition) still fills the chip
    const bool want_w = dw != nullptr || db != nullptr;          // both NULL: dx only (no read of x, no partial rows, no reduction launch)
    MDVIT_CHECK_ARG(want_w || dx != nullptr, MDVIT_E_SHAPE, "rowdot_bwd: nothing to produce");
    if (want_w) MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nblk, K + 1, "rowdot_bwd");
    dim3 grid(nblk);
    const int C = K;
    float* part = want_w ? (float*)ws : nullptr;
    LN_DISPATCH(rowdot_bwd_kernel, C, x, (long)ldx, w, dy, dx, (long)lddx, part, db != nullptr, M, K);
    MDVIT_LAUNCH_CHECK();
    if (!want_w) return MDVIT_OK;
    return mdvit_reduce_partials(part, nblk, (long)K + 1, K, dw, 1, db, 0, s);
}
