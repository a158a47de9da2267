// 3x3 convolutions and bilinear resampling on NHWC fp32 (HBM-bound; lanes run along channels so
// every global access is a contiguous 16 B/lane float4, halo re-reads are served by L1/L2).
#include "common.h"
#include "conv_tile.h"

namespace {

__device__ __forceinline__ float4 f4_fma(float4 a, float4 w, float4 acc) {
    acc.x = fmaf(a.x, w.x, acc.x); acc.y = fmaf(a.y, w.y, acc.y); acc.z = fmaf(a.z, w.z, acc.z); acc.w = fmaf(a.w, w.w, acc.w);
    return acc;
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ---- depthwise 3x3, pad 1, stride 1|2 --------------------------------------------------------
// thread = (output token, channel quad).  w [C,1,3,3] -> LDS as [9][C] once per block.
__global__ __launch_bounds__(256) void dwconv3x3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int B, int Hi, int Wi, int C, int stride, int add_input) {
    extern __shared__ float s_w[];   // [9][C]
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) { const int c = i / 9, t = i % 9; s_w[t * C + c] = w[i]; }
    __syncthreads();
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1, QC = C >> 2;
    const long total = (long)B * Ho * Wo * QC;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % QC) * 4;
        long tkn = e / QC;
        const int wo = (int)(tkn % Wo); tkn /= Wo;
        const int ho = (int)(tkn % Ho);
        const int b = (int)(tkn / Ho);
        float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * stride + kh - 1, hc = min(max(hi, 0), Hi - 1);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * stride + kw - 1, wc = min(max(wi, 0), Wi - 1);
                const float m = (hi == hc && wi == wc) ? 1.f : 0.f;
                const float4 xv = *reinterpret_cast<const float4*>(x + (((long)b * Hi + hc) * Wi + wc) * C + c);
                const float4 wv = *reinterpret_cast<const float4*>(&s_w[(kh * 3 + kw) * C + c]);
                acc = f4_fma(xv, make_float4(wv.x * m, wv.y * m, wv.z * m, wv.w * m), acc);
            }
        }
        if (add_input) acc = f4_add(acc, *reinterpret_cast<const float4*>(x + (((long)b * Hi + ho) * Wi + wo) * C + c));
        *reinterpret_cast<float4*>(y + (((long)b * Ho + ho) * Wo + wo) * C + c) = acc;
    }
}

// dgrad: dx[b,hi,wi,c] = sum_{kh,kw} w[c,kh,kw] * dy[b,ho,wo,c], ho*stride + kh - 1 == hi  (+ dy if add_input)
__global__ __launch_bounds__(256) void dwconv3x3_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, int B, int Hi, int Wi, int C, int stride, int add_input) {
    extern __shared__ float s_w[];
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) { const int c = i / 9, t = i % 9; s_w[t * C + c] = w[i]; }
    __syncthreads();
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1, QC = C >> 2;
    const long total = (long)B * Hi * Wi * QC;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % QC) * 4;
        long tkn = e / QC;
        const int wi = (int)(tkn % Wi); tkn /= Wi;
        const int hi = (int)(tkn % Hi);
        const int b = (int)(tkn / Hi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hn = hi - kh + 1;
            if (hn < 0 || (hn % stride) != 0) continue;
            const int ho = hn / stride;
            if (ho >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wn = wi - kw + 1;
                if (wn < 0 || (wn % stride) != 0) continue;
                const int wo = wn / stride;
                if (wo >= Wo) continue;
                const float4 g = *reinterpret_cast<const float4*>(dy + (((long)b * Ho + ho) * Wo + wo) * C + c);
                const float4 wv = *reinterpret_cast<const float4*>(&s_w[(kh * 3 + kw) * C + c]);
                acc = f4_fma(g, wv, acc);
            }
        }
        if (add_input) acc = f4_add(acc, *reinterpret_cast<const float4*>(dy + (((long)b * Hi + hi) * Wi + wi) * C + c));
        *reinterpret_cast<float4*>(dx + (((long)b * Hi + hi) * Wi + wi) * C + c) = acc;
    }
}

// wgrad: dw[c,kh,kw] = sum_{b,ho,wo} dy[b,ho,wo,c] * x[b,ho*s+kh-1,wo*s+kw-1,c]; dbias[c] = sum dy.
// Block = (C/4 quads) x (256/(C/4) token lanes) over a contiguous chunk of output tokens; register
// partials -> LDS atomics -> one global atomic per (c,tap) per block.
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ part,
                                                              int B, int Hi, int Wi, int C, int stride, int tokens_per_block) {
    extern __shared__ float s_acc[];   // [C][10]
    for (int i = threadIdx.x; i < C * 10; i += blockDim.x) s_acc[i] = 0.f;
    __syncthreads();
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1, QC = C >> 2;
    const long ntok = (long)B * Ho * Wo;
    const long t_beg = (long)blockIdx.x * tokens_per_block, t_end = min(ntok, t_beg + tokens_per_block);
    const int nquads_blk = min(QC, 256);
    const int tl = threadIdx.x / nquads_blk, ntl = blockDim.x / nquads_blk;
    for (int q = threadIdx.x % nquads_blk; q < QC; q += nquads_blk) {
        const int c = q * 4;
        float4 acc[9];
        float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tl < ntl) {
            for (long tk = t_beg + tl; tk < t_end; tk += ntl) {
                long r = tk;
                const int wo = (int)(r % Wo); r /= Wo;
                const int ho = (int)(r % Ho);
                const int b = (int)(r / Ho);
                const float4 g = *reinterpret_cast<const float4*>(dy + tk * C + c);
                accb = f4_add(accb, g);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int hi = ho * stride + kh - 1, hc = min(max(hi, 0), Hi - 1);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int wi = wo * stride + kw - 1, wc = min(max(wi, 0), Wi - 1);
                        const float m = (hi == hc && wi == wc) ? 1.f : 0.f;
                        const float4 xv = *reinterpret_cast<const float4*>(x + (((long)b * Hi + hc) * Wi + wc) * C + c);
                        acc[kh * 3 + kw] = f4_fma(make_float4(g.x * m, g.y * m, g.z * m, g.w * m), xv, acc[kh * 3 + kw]);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            atomicAdd(&s_acc[(c + 0) * 10 + t], acc[t].x); atomicAdd(&s_acc[(c + 1) * 10 + t], acc[t].y);
            atomicAdd(&s_acc[(c + 2) * 10 + t], acc[t].z); atomicAdd(&s_acc[(c + 3) * 10 + t], acc[t].w);
        }
        atomicAdd(&s_acc[(c + 0) * 10 + 9], accb.x); atomicAdd(&s_acc[(c + 1) * 10 + 9], accb.y);
        atomicAdd(&s_acc[(c + 2) * 10 + 9], accb.z); atomicAdd(&s_acc[(c + 3) * 10 + 9], accb.w);
    }
    __syncthreads();
    float* row = part + (long)blockIdx.x * C * 10;       // [dw (C*9) | dbias (C)], summed over workgroups by mdvit_reduce_partials
    for (int i = threadIdx.x; i < C * 10; i += blockDim.x) {
        const int c = i / 10, t = i % 10;
        row[t < 9 ? c * 9 + t : C * 9 + c] = s_acc[i];
    }
}

// ---- grouped conv on the virtual concat(skip, up): out channel g <- concat channels 2g, 2g+1 ---
// thread = (token, output-channel pair): reads one float4 = concat channels [4p, 4p+4) -> outputs 2p, 2p+1
__device__ __forceinline__ const float* cat_ptr(const float* skip, const float* up, int C, long tok, int cc) {
    // cc: concat channel (multiple of 4) in [0, 2C)
    return cc < C ? skip + tok * C + cc : up + tok * C + (cc - C);
}

__global__ __launch_bounds__(256) void gconv2_fwd_kernel(const float* __restrict__ skip, const float* __restrict__ up,
                                                         const float* __restrict__ w, float* __restrict__ y, int B, int H, int W, int C) {
    extern __shared__ float s_w[];   // [9][2C]: tap-major, concat-channel minor
    for (int i = threadIdx.x; i < 18 * C; i += blockDim.x) {
        const int g = i / 18, r = i % 18, j = r / 9, t = r % 9;     // w[g][j][t]
        s_w[t * 2 * C + 2 * g + j] = w[i];
    }
    __syncthreads();
    const int PC = C >> 1;
    const long total = (long)B * H * W * PC;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int p = (int)(e % PC);
        long tkn = e / PC;
        const int wo = (int)(tkn % W); tkn /= W;
        const int ho = (int)(tkn % H);
        const int b = (int)(tkn / H);
        const int cc = 4 * p;
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho + kh - 1;
            if (hi < 0 || hi >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo + kw - 1;
                if (wi < 0 || wi >= W) continue;
                const float4 xv = *reinterpret_cast<const float4*>(cat_ptr(skip, up, C, ((long)b * H + hi) * W + wi, cc));
                const float4 wv = *reinterpret_cast<const float4*>(&s_w[(kh * 3 + kw) * 2 * C + cc]);
                o0 = fmaf(xv.x, wv.x, fmaf(xv.y, wv.y, o0));
                o1 = fmaf(xv.z, wv.z, fmaf(xv.w, wv.w, o1));
            }
        }
        *reinterpret_cast<float2*>(y + (((long)b * H + ho) * W + wo) * C + 2 * p) = make_float2(o0, o1);
    }
}

// dgrad wrt the concat: d cat[b,hi,wi,2g+j] = sum_taps w[g,j,kh,kw] dy[b,hi-kh+1,wi-kw+1,g]
__global__ __launch_bounds__(256) void gconv2_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           float* __restrict__ dskip, float* __restrict__ dup, int B, int H, int W, int C) {
    extern __shared__ float s_w[];
    for (int i = threadIdx.x; i < 18 * C; i += blockDim.x) {
        const int g = i / 18, r = i % 18, j = r / 9, t = r % 9;
        s_w[t * 2 * C + 2 * g + j] = w[i];
    }
    __syncthreads();
    const int PC = C >> 1;
    const long total = (long)B * H * W * PC;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int p = (int)(e % PC);
        long tkn = e / PC;
        const int wi = (int)(tkn % W); tkn /= W;
        const int hi = (int)(tkn % H);
        const int b = (int)(tkn / H);
        const int cc = 4 * p;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ho = hi - kh + 1;
            if (ho < 0 || ho >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wo = wi - kw + 1;
                if (wo < 0 || wo >= W) continue;
                const float2 g = *reinterpret_cast<const float2*>(dy + (((long)b * H + ho) * W + wo) * C + 2 * p);
                const float4 wv = *reinterpret_cast<const float4*>(&s_w[(kh * 3 + kw) * 2 * C + cc]);
                acc.x = fmaf(g.x, wv.x, acc.x); acc.y = fmaf(g.x, wv.y, acc.y);
                acc.z = fmaf(g.y, wv.z, acc.z); acc.w = fmaf(g.y, wv.w, acc.w);
            }
        }
        const long tok = ((long)b * H + hi) * W + wi;
        float* dst = cc < C ? dskip + tok * C + cc : dup + tok * C + (cc - C);
        *reinterpret_cast<float4*>(dst) = acc;
    }
}

// The same two passes on LDS tiles (round 5; C % 32 == 0): w [C][2][3][3] IS a depthwise filter bank over the 2 C concat channels, so the forward is conv_tile.h's
// 8 x 16 x 32 tile pass over each source + the sum of channel pairs, the data gradient the flipped-tap pass with one dy channel feeding its group's two concat channels.
// The element-per-thread kernels above issue nine loads under nine branches per output: 85 / 73 us per launch at the four decoder stages of a bs=4 step, 14 % of their
// bytes' HBM time, on the forward's single stream and on both backward sweeps.  blockIdx.y = 32-channel block of the CONCAT axis.
template <bool DGRAD>
__global__ __launch_bounds__(256) void gconv2_tile_kernel(const float* __restrict__ a0, const float* __restrict__ a1, const float* __restrict__ w,
                                                          float* __restrict__ o0, float* __restrict__ o1, int H, int W, int C, int tiles_w) {
    __shared__ __attribute__((aligned(16))) float sx[(CT_TH + 2) * (CT_TW + 2) * CT_CL];
    __shared__ float sw[CT_CL * 9];
    const int cb = blockIdx.y, half = C / CT_CL;
    const bool first = cb < half;
    const int lb = first ? cb : cb - half;                     // block inside its source
    const float* wh = w + (first ? 0L : (long)C * 9);
    if (!DGRAD) conv_tile_body<3, false, 1, true>(sx, sw, lb, first ? a0 : a1, (long)C, 0, wh, nullptr, o0, (long)C, first ? 0 : C / 2, H, W, C, tiles_w, 0);
    else conv_tile_body<3, true, 2, false>(sx, sw, lb, a0 + (first ? 0 : C / 2), (long)C, 0, wh, nullptr, first ? o0 : o1, (long)C, 0, H, W, C, tiles_w, 0);
}

// wgrad: dw[g,j,kh,kw] = sum dy[b,ho,wo,g] * cat[b,ho+kh-1,wo+kw-1,2g+j]
__global__ __launch_bounds__(256) void gconv2_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ skip,
                                                           const float* __restrict__ up, float* __restrict__ part,
                                                           int B, int H, int W, int C, int tokens_per_block) {
    extern __shared__ float s_acc[];   // [C][18]
    for (int i = threadIdx.x; i < C * 18; i += blockDim.x) s_acc[i] = 0.f;
    __syncthreads();
    const int PC = C >> 1;
    const long ntok = (long)B * H * W;
    const long t_beg = (long)blockIdx.x * tokens_per_block, t_end = min(ntok, t_beg + tokens_per_block);
    const int np_blk = min(PC, 256);
    const int tl = threadIdx.x / np_blk, ntl = blockDim.x / np_blk;
    for (int p = threadIdx.x % np_blk; p < PC; p += np_blk) {
        const int cc = 4 * p;
        float4 acc[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tl < ntl) {
            for (long tk = t_beg + tl; tk < t_end; tk += ntl) {
                long r = tk;
                const int wo = (int)(r % W); r /= W;
                const int ho = (int)(r % H);
                const int b = (int)(r / H);
                const float2 g = *reinterpret_cast<const float2*>(dy + tk * C + 2 * p);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int hi = ho + kh - 1;
                    if (hi < 0 || hi >= H) continue;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int wi = wo + kw - 1;
                        if (wi < 0 || wi >= W) continue;
                        const float4 xv = *reinterpret_cast<const float4*>(cat_ptr(skip, up, C, ((long)b * H + hi) * W + wi, cc));
                        float4& a = acc[kh * 3 + kw];
                        a.x = fmaf(g.x, xv.x, a.x); a.y = fmaf(g.x, xv.y, a.y);
                        a.z = fmaf(g.y, xv.z, a.z); a.w = fmaf(g.y, xv.w, a.w);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {   // outputs 2p (j=0,1) and 2p+1 (j=0,1)
            atomicAdd(&s_acc[(2 * p) * 18 + t], acc[t].x);     atomicAdd(&s_acc[(2 * p) * 18 + 9 + t], acc[t].y);
            atomicAdd(&s_acc[(2 * p + 1) * 18 + t], acc[t].z); atomicAdd(&s_acc[(2 * p + 1) * 18 + 9 + t], acc[t].w);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 18; i += blockDim.x) part[(long)blockIdx.x * C * 18 + i] = s_acc[i];
}

// ---- im2col / col2im for dense 3x3 pad 1; column order (cin, kh, kw) ---------------------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, float* __restrict__ col,
                                                        int B, int Hi, int Wi, int Cin, int stride, int dil) {
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    const long total = (long)B * Ho * Wo * Cin;
    const int K = Cin * 9;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % Cin);
        long m = e / Cin;
        long r = m;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float* dst = col + m * K + c * 9;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * stride + (kh - 1) * dil;             // padding = dilation (a "same" 3x3 at stride 1)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * stride + (kw - 1) * dil;
                const bool ok = hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;
                dst[kh * 3 + kw] = ok ? x[(((long)b * Hi + hi) * Wi + wi) * Cin + c] : 0.f;
            }
        }
    }
}

__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dcol, float* __restrict__ dx,
                                                        int B, int Hi, int Wi, int Cin, int stride, int dil) {
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    const long total = (long)B * Hi * Wi * Cin;
    const int K = Cin * 9;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % Cin);
        long r = e / Cin;
        const int wi = (int)(r % Wi); r /= Wi;
        const int hi = (int)(r % Hi);
        const int b = (int)(r / Hi);
        float acc = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hn = hi - (kh - 1) * dil;
            if (hn < 0 || (hn % stride) != 0) continue;
            const int ho = hn / stride;
            if (ho >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wn = wi - (kw - 1) * dil;
                if (wn < 0 || (wn % stride) != 0) continue;
                const int wo = wn / stride;
                if (wo >= Wo) continue;
                acc += dcol[(((long)b * Ho + ho) * Wo + wo) * K + c * 9 + kh * 3 + kw];
            }
        }
        dx[e] = acc;
    }
}

// ---- stem.0: NCHW image -> NHWC, 3x3 s2 p1; thread = (output pixel, 4 output channels) ----------
template <int CIN>
__global__ __launch_bounds__(256) void stemconv_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                           float* __restrict__ y, int B, int H, int W, int Cout) {
    extern __shared__ float s_w[];   // [CIN*9][Cout]
    for (int i = threadIdx.x; i < Cout * CIN * 9; i += blockDim.x) { const int co = i / (CIN * 9), k = i % (CIN * 9); s_w[k * Cout + co] = w[i]; }
    __syncthreads();
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, QC = Cout >> 2;
    const long total = (long)B * Ho * Wo * QC;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int co = (int)(e % QC) * 4;
        long r = e / QC;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hi = 2 * ho + kh - 1;
                if (hi < 0 || hi >= H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int wi = 2 * wo + kw - 1;
                    if (wi < 0 || wi >= W) continue;
                    const float xv = img[(((long)b * CIN + ci) * H + hi) * W + wi];
                    const float4 wv = *reinterpret_cast<const float4*>(&s_w[(ci * 9 + kh * 3 + kw) * Cout + co]);
                    acc.x = fmaf(xv, wv.x, acc.x); acc.y = fmaf(xv, wv.y, acc.y); acc.z = fmaf(xv, wv.z, acc.z); acc.w = fmaf(xv, wv.w, acc.w);
                }
            }
        *reinterpret_cast<float4*>(y + (((long)b * Ho + ho) * Wo + wo) * Cout + co) = acc;
    }
}

// Cout == 32 (the stem of every model here) on the matrix pipe (round 5): y[px][co] = sum_k tap_k(px) w[co][k] as v_mfma_f32_32x32x2_f32 over k pairs -- exact fp32
// products, fp32 accumulate in ascending k.  A wave owns 32 output pixels per pass: lane (pixel l & 31, k parity l >> 5) gathers ONE image value per k pair (14 loads for
// the 27 taps; the kernel above issues 27 loads from each of a pixel's eight channel-quad threads: 150 us for the 16 images of a bs=4 step, alone on the GPU at the top of
// the forward), the weights sit in 14 registers per lane, and the accumulator's rows leave as 128-byte channel rows.  (A pixel per thread with the 32 channels in registers
// and the weights as LDS broadcasts was built first: 218 us -- 216 ds_read_b128 per pixel.)
template <int CIN>
__global__ __launch_bounds__(256) void stemconv_fwd32_kernel(const float* __restrict__ img, const float* __restrict__ w, float* __restrict__ y, int B, int H, int W) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int Cout = 32, KK = CIN * 9, KP = (KK + 1) / 2;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo;
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    float wk[KP];                                    // B operand: lane (co = l31, k = 2 s + lhi)
    int toff[KP];                                    // the tap's (ci, kh, kw) packed: ci << 4 | kh << 2 | kw; -1 past the last tap
#pragma unroll
    for (int s = 0; s < KP; ++s) {
        const int k = 2 * s + lhi;
        wk[s] = k < KK ? w[l31 * KK + k] : 0.f;
        toff[s] = k < KK ? ((k / 9) << 4 | ((k % 9) / 3) << 2 | (k % 3)) : -1;
    }
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6), wid = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (long p0 = wid * 32; p0 < total; p0 += nwaves * 32) {
        const long px = min(p0 + l31, total - 1);
        long r = px;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float xv[KP];
#pragma unroll
        for (int s = 0; s < KP; ++s) {
            const int t = toff[s], ci = max(t, 0) >> 4, kh = (t >> 2) & 3, kw = t & 3;
            const int hi = 2 * ho + kh - 1, wi = 2 * wo + kw - 1;
            const bool ok = t >= 0 && hi >= 0 && hi < H && wi >= 0 && wi < W;
            const float v = img[(((long)b * CIN + ci) * H + min(max(hi, 0), H - 1)) * W + min(max(wi, 0), W - 1)];
            xv[s] = ok ? v : 0.f;
        }
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int s = 0; s < KP; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[s], wk[s], acc, 0, 0, 0);
        // D[i = pixel][j = co]: lane holds column co = l31, rows i = (q & 3) + 8 (q >> 2) + 4 lhi
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long po = p0 + (q & 3) + 8 * (q >> 2) + 4 * lhi;
            if (po < total) y[po * Cout + l31] = acc[q];
        }
    }
}

// dw[co,ci,kh,kw] = sum_{b,ho,wo} dy[b,ho,wo,co] * img[b,ci,2ho+kh-1,2wo+kw-1]
// thread = (co, pixel lane); 27 register accumulators; LDS atomics -> global atomics.
template <int CIN>
__global__ __launch_bounds__(256) void stemconv_wgrad_kernel(const float* __restrict__ img, const float* __restrict__ dy,
                                                             float* __restrict__ part, int B, int H, int W, int Cout, int pix_per_block) {
    extern __shared__ float s_acc[];   // [Cout][CIN*9]
    constexpr int KK = CIN * 9;
    for (int i = threadIdx.x; i < Cout * KK; i += blockDim.x) s_acc[i] = 0.f;
    __syncthreads();
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long npix = (long)B * Ho * Wo;
    const long p_beg = (long)blockIdx.x * pix_per_block, p_end = min(npix, p_beg + pix_per_block);
    const int nco = min(Cout, 256), pl = threadIdx.x / nco, npl = blockDim.x / nco;
    for (int co = threadIdx.x % nco; co < Cout; co += nco) {
        float acc[KK];
#pragma unroll
        for (int k = 0; k < KK; ++k) acc[k] = 0.f;
        if (pl < npl) {
            for (long px = p_beg + pl; px < p_end; px += npl) {
                long r = px;
                const int wo = (int)(r % Wo); r /= Wo;
                const int ho = (int)(r % Ho);
                const int b = (int)(r / Ho);
                const float g = dy[px * Cout + co];
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int hi = 2 * ho + kh - 1;
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int wi = 2 * wo + kw - 1;
                            const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
                            const float xv = ok ? img[(((long)b * CIN + ci) * H + hi) * W + wi] : 0.f;
                            acc[ci * 9 + kh * 3 + kw] = fmaf(g, xv, acc[ci * 9 + kh * 3 + kw]);
                        }
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) atomicAdd(&s_acc[co * KK + k], acc[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Cout * KK; i += blockDim.x) part[(long)blockIdx.x * Cout * KK + i] = s_acc[i];
}

// The same sums on the matrix pipe (round 5; Cout % 32 == 0): D[co][tap] += dy[px][co] * tap(px) as v_mfma_f32_32x32x2_f32 over PAIRS of output pixels -- exact fp32
// products, fp32 accumulate.  The kernel above issues 27 image loads per output pixel from every one of its 32 channel lanes (two addresses per wave-instruction:
// 396 us for the 16 x 512 x 512 images of a bs=4 step, 215 MB of HBM traffic at 0.5 TB/s).  Here lane (pixel l >> 5, co l & 31) loads ONE dy value (a wave reads two
// 128-byte rows) and lane (pixel l >> 5, tap l & 31) gathers ONE image value per pixel pair; a wave walks its pixels with eight pairs in flight.  The four waves'
// accumulators are added in wave order through LDS: one partial row [Cout][27] per workgroup, summed by mdvit_reduce_partials as before (deterministic).
template <int CIN>
__global__ __launch_bounds__(256) void stemconv_wgrad_mfma_kernel(const float* __restrict__ img, const float* __restrict__ dy,
                                                                  float* __restrict__ part, int B, int H, int W, int Cout, int pix_per_block) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int KK = CIN * 9;
    static_assert(KK <= 32, "one 32-wide tap block");
    __shared__ float s_acc[4][32 * 33];
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long npix = (long)B * Ho * Wo;
    const long p_beg = (long)blockIdx.x * pix_per_block, p_end = min(npix, p_beg + pix_per_block);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int tap = min(l31, KK - 1), ci = tap / 9, kh = (tap % 9) / 3, kw = tap % 3;
    const bool tap_ok = l31 < KK;
    constexpr int UN = 8;                            // pixel pairs in flight per wave
    for (int cb = 0; cb < Cout; cb += 32) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (long p0 = p_beg + wave * 2 * UN; p0 < p_end; p0 += 4 * 2 * UN) {
            float a[UN], x[UN];
            // (image, row, column) of p0 once per pass, on the scalar unit (p0 is wave-uniform); the 16 pixels of the pass follow by carries
            const long p0s = ((long)__builtin_amdgcn_readfirstlane((int)(p0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)p0);
            const int wo0 = (int)(p0s % Wo), ho0 = (int)((p0s / Wo) % Ho), b0 = (int)(p0s / Wo / Ho);
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long px = p0 + 2 * u + lhi;
                const bool pok = px < p_end;
                const long pc = pok ? px : p_end - 1;
                int wo = wo0 + 2 * u + lhi - (int)(px - pc), ho = ho0, b = b0;
                while (wo >= Wo) { wo -= Wo; ++ho; }
                while (ho >= Ho) { ho -= Ho; ++b; }
                const int hi = 2 * ho + kh - 1, wi = 2 * wo + kw - 1;
                const bool ok = pok && tap_ok && hi >= 0 && hi < H && wi >= 0 && wi < W;
                const float g = dy[pc * Cout + cb + l31];
                const float v = img[(((long)b * CIN + ci) * H + min(max(hi, 0), H - 1)) * W + min(max(wi, 0), W - 1)];
                a[u] = pok ? g : 0.f;
                x[u] = ok ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], x[u], acc, 0, 0, 0);
        }
        // D[i = co][j = tap]: lane holds column j = l31, rows i = (r & 3) + 8 (r >> 2) + 4 lhi
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[wave][((r & 3) + 8 * (r >> 2) + 4 * lhi) * 33 + l31] = acc[r];
        __syncthreads();
        for (int i = threadIdx.x; i < 32 * KK; i += 256) {
            const int co = i / KK, k = i % KK;
            const float t = ((s_acc[0][co * 33 + k] + s_acc[1][co * 33 + k]) + s_acc[2][co * 33 + k]) + s_acc[3][co * 33 + k];
            part[(long)blockIdx.x * Cout * KK + (cb + co) * KK + k] = t;
        }
    }
}

// ---- bilinear, align_corners=False (ATen upsample_bilinear2d index rule) ------------------------
__device__ __forceinline__ void bilin_src(int o, int in_size, float scale, int& i0, int& i1, float& l1) {
    float src = scale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ x, const float* base_, float* y,
                                                           int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    const bool vec = (C & 3) == 0;
    const int QC = vec ? C >> 2 : C;
    const long total = (long)B * Ho * Wo * QC;
    // (adjacent output pixels share their source taps: neighbouring workgroups on one XCD)
    for (long e = (long)mdvit_xcd_logical_block() * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int q = (int)(e % QC);
        long r = e / QC;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        int h0, h1, w0, w1; float lh, lw;
        bilin_src(ho, Hi, sh, h0, h1, lh);
        bilin_src(wo, Wi, sw, w0, w1, lw);
        const float c00 = (1.f - lh) * (1.f - lw), c01 = (1.f - lh) * lw, c10 = lh * (1.f - lw), c11 = lh * lw;
        const long base = (long)b * Hi * Wi;
        if (vec) {
            const int c = q * 4;
            const float4 v00 = *reinterpret_cast<const float4*>(x + (base + (long)h0 * Wi + w0) * C + c);
            const float4 v01 = *reinterpret_cast<const float4*>(x + (base + (long)h0 * Wi + w1) * C + c);
            const float4 v10 = *reinterpret_cast<const float4*>(x + (base + (long)h1 * Wi + w0) * C + c);
            const float4 v11 = *reinterpret_cast<const float4*>(x + (base + (long)h1 * Wi + w1) * C + c);
            float4 o;
            o.x = c00 * v00.x + c01 * v01.x + c10 * v10.x + c11 * v11.x;
            o.y = c00 * v00.y + c01 * v01.y + c10 * v10.y + c11 * v11.y;
            o.z = c00 * v00.z + c01 * v01.z + c10 * v10.z + c11 * v11.z;
            o.w = c00 * v00.w + c01 * v01.w + c10 * v10.w + c11 * v11.w;
            const long oi = (((long)b * Ho + ho) * Wo + wo) * C + c;
            if (base_) o = f4_add(o, *reinterpret_cast<const float4*>(base_ + oi));        // y = base + up (base may be y itself)
            *reinterpret_cast<float4*>(y + oi) = o;
        } else {
            const float v = c00 * x[(base + (long)h0 * Wi + w0) * C + q] + c01 * x[(base + (long)h0 * Wi + w1) * C + q] +
                            c10 * x[(base + (long)h1 * Wi + w0) * C + q] + c11 * x[(base + (long)h1 * Wi + w1) * C + q];
            const long oi = (((long)b * Ho + ho) * Wo + wo) * C + q;
            y[oi] = base_ ? base_[oi] + v : v;
        }
    }
}

// adjoint in gather form: for every input pixel, visit the output pixels whose 2-tap stencil can touch it.
__device__ __forceinline__ void out_range(int i, int in_size, int out_size, int& lo, int& hi) {
    // outputs o with floor(src(o)) in {i-1, i}; exact weights are re-derived per o
    if (out_size % in_size == 0 && ((out_size / in_size) & 1) == 0) {
        // even integer scale s (every resize of this model: x2 / x4 / x8): src(o) = (o + 0.5) / s - 0.5, so input i is touched exactly by
        // o in [s i - s/2, s i + 3 s / 2) -- 2 s taps instead of the conservative 3 s + 2 (the clamped border outputs fall inside too)
        const int s = out_size / in_size;
        lo = s * i - s / 2; hi = s * i + 3 * s / 2;
        if (lo < 0) lo = 0;
        if (hi > out_size) hi = out_size;
        return;
    }
    const float inv = (float)out_size / (float)in_size;
    lo = (int)floorf(((float)i - 1.0f) * inv) - 1;
    hi = (int)ceilf(((float)i + 2.0f) * inv) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size) hi = out_size;
}

// Separable adjoint (bilinear weights factor as wh * ww): pass W folds the output columns into the input columns
// (tmp [B,Ho,Wi,C]), pass H folds the rows.  2*(2s+1) taps per result instead of (2s+1)^2 (s = scale: the peer heads
// upsample 512-channel features x2/x4/x8), every tap load unconditional (weight 0 when it does not touch the pixel).
template <bool ALONG_W>
__global__ __launch_bounds__(256) void upsample_bwd_pass_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                long outer, int n_in, int n_out, long inner_q, int vec) {
    // src [outer][n_out][inner], dst [outer][n_in][inner]   (ALONG_W: outer = B*Ho, inner = C;  else outer = B, inner = Wi*C)
    const float sc = (float)n_in / (float)n_out;
    const long total = outer * n_in * inner_q;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long q = e % inner_q;
        long r = e / inner_q;
        const int i = (int)(r % n_in);
        const long o = r / n_in;
        int lo, hi;
        out_range(i, n_in, n_out, lo, hi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const long inner = vec ? inner_q * 4 : inner_q;
        const float* base = src + o * n_out * inner + (vec ? q * 4 : q);
#pragma unroll 4
        for (int t = lo; t < hi; ++t) {
            int i0, i1; float l;
            bilin_src(t, n_in, sc, i0, i1, l);
            const float wgt = (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
            if (vec) {
                const float4 g = *reinterpret_cast<const float4*>(base + (long)t * inner);
                acc.x = fmaf(wgt, g.x, acc.x); acc.y = fmaf(wgt, g.y, acc.y); acc.z = fmaf(wgt, g.z, acc.z); acc.w = fmaf(wgt, g.w, acc.w);
            } else {
                acc.x = fmaf(wgt, base[(long)t * inner], acc.x);
            }
        }
        float* d = dst + (o * n_in + i) * inner + (vec ? q * 4 : q);
        if (vec) *reinterpret_cast<float4*>(d) = acc;
        else *d = acc.x;
    }
}

// ---- several bilinear sources summed onto one base in ONE pass (the peer heads' fuse: sum_q upsample(P_q), Decoders.py:320-331 after the
// weight composition of decode.py).  Forward: y = base + sum_i resize(x_i); chained single-source calls read and write the [B,Ho,Wo,C] sum once
// per source.  Backward, width pass: every source's [B*Ho][Wi_i][C] fold of dy in one launch, ordered so that the three folds of a dy row run
// together (the row comes from HBM once, from L2 afterwards); the height passes stay per source (their inputs are 2-8x smaller).
struct UpMulti {
    const float* x[3]; float* d[3];
    int Hi[3], Wi[3];
    int n;
};

__global__ __launch_bounds__(256) void upsample_multi_fwd_kernel(UpMulti p, const float* base_, float* __restrict__ y, int B, int Ho, int Wo, int C) {
    const int QC = C >> 2;
    // Work unit = 1024 consecutive channel quads of ONE output row (four per thread): the row / chunk split is a wave-uniform division (scalar unit), the
    // vertical taps of the row are computed once per unit, and what is left per element is one 32-bit division -- the flat 64-bit index arithmetic per element
    // (three long divisions) is gone; 129 -> 125 us per peer head at 4 x 128 x 128 x 512: the kernel is bound by the texture-address path -- 12 tap loads (L1 / L2 hits)
    // per 16 bytes stored, ~48 us of load issue per CU under the 62 us of HBM time for its 310 MB -- not by index arithmetic, and streaming (nontemporal) base / y
    // accesses changed nothing; the next step would be source tiles in LDS.
    // (2-8 horizontally adjacent output pixels read the same source pixels: consecutive logical blocks -- the chunks of a row -- sit on one XCD; PMC FETCH 283 MB
    //  per launch before that, for 178 MB of base + sources)
    const int row_items = Wo * QC;
    const int cpr = (row_items + 1023) >> 10;
    const long nunits = (long)B * Ho * cpr;
    for (long u = mdvit_xcd_logical_block(); u < nunits; u += gridDim.x) {
        const long row = u / cpr;
        const int k = (int)(u - row * cpr);
        const int ho = (int)(row % Ho), b = (int)(row / Ho);
        int h0[3], h1[3]; float lh[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            h0[i] = h1[i] = 0; lh[i] = 0.f;
            if (i < p.n) bilin_src(ho, p.Hi[i], (float)p.Hi[i] / (float)Ho, h0[i], h1[i], lh[i]);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = (k << 10) + it * 256 + (int)threadIdx.x;
            if (e >= row_items) break;
            const int wo = e / QC, c = (e - wo * QC) * 4;
            const long oi = ((row * Wo) + wo) * C + c;
            float4 o = base_ ? *reinterpret_cast<const float4*>(base_ + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i >= p.n) break;
                const int Hi = p.Hi[i], Wi = p.Wi[i];
                int w0, w1; float lw;
                bilin_src(wo, Wi, (float)Wi / (float)Wo, w0, w1, lw);
                const float c00 = (1.f - lh[i]) * (1.f - lw), c01 = (1.f - lh[i]) * lw, c10 = lh[i] * (1.f - lw), c11 = lh[i] * lw;
                const float* x = p.x[i] + (long)b * Hi * Wi * C + c;
                const float4 v00 = *reinterpret_cast<const float4*>(x + ((long)h0[i] * Wi + w0) * C), v01 = *reinterpret_cast<const float4*>(x + ((long)h0[i] * Wi + w1) * C);
                const float4 v10 = *reinterpret_cast<const float4*>(x + ((long)h1[i] * Wi + w0) * C), v11 = *reinterpret_cast<const float4*>(x + ((long)h1[i] * Wi + w1) * C);
                // the same four-tap expression as upsample_fwd_kernel, then added to the running sum: equal to the chained calls bit for bit
                float4 uu;
                uu.x = c00 * v00.x + c01 * v01.x + c10 * v10.x + c11 * v11.x;
                uu.y = c00 * v00.y + c01 * v01.y + c10 * v10.y + c11 * v11.y;
                uu.z = c00 * v00.z + c01 * v01.z + c10 * v10.z + c11 * v11.z;
                uu.w = c00 * v00.w + c01 * v01.w + c10 * v10.w + c11 * v11.w;
                o = (i == 0 && !base_) ? uu : f4_add(uu, o);
            }
            *reinterpret_cast<float4*>(y + oi) = o;
        }
    }
}

// The same sum with the sources' pixels staged in LDS: a workgroup owns an 8 x 16 pixel tile x 128 channels of the output; the source patch under the tile
// (rows h0(first row) .. h1(last row), columns likewise) is read from HBM / L2 ONCE per workgroup -- 96 pixels in all for upscale factors 2 / 4 / 8 -- and the four
// taps of every output pixel come out of LDS.  The flat kernel issues 12 tap loads (L1 / L2 hits, but texture-address work) per 16 bytes it stores and is bound by
// that path (125 us per peer head for 310 MB: 2.5 TB/s); here a workgroup moves 32 KB of base + 24 KB of sources + 32 KB of output.  Same tap expression on the same
// values.  Eligibility (host): Ho % 8 == 0, Wo % 16 == 0, C % 128 == 0, integer upscale factors >= 2 whose patches fit the 96-pixel pool.
constexpr int UT_H = 8, UT_W = 16, UT_C = 128, UT_POOL = 96;
__global__ __launch_bounds__(256) void upsample_multi_tile_kernel(UpMulti p, const float* __restrict__ base_, float* __restrict__ y, int Ho, int Wo, int C) {
    __shared__ __attribute__((aligned(16))) float s_src[UT_POOL * UT_C];
    const int tiles_w = Wo / UT_W;
    const int ty = blockIdx.x / tiles_w, tx = blockIdx.x - ty * tiles_w;
    const int ho0 = ty * UT_H, wo0 = tx * UT_W, c0 = blockIdx.y * UT_C, b = blockIdx.z;
    int r0[3], q0[3], pw[3], off[3];
    {
        int used = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            r0[i] = q0[i] = pw[i] = off[i] = 0;
            if (i >= p.n) continue;
            const int Hi = p.Hi[i], Wi = p.Wi[i];
            int a0, a1, b0, b1; float l;
            bilin_src(ho0, Hi, (float)Hi / (float)Ho, a0, b0, l);
            bilin_src(ho0 + UT_H - 1, Hi, (float)Hi / (float)Ho, a1, b1, l);
            const int ph = b1 - a0 + 1;
            r0[i] = a0;
            bilin_src(wo0, Wi, (float)Wi / (float)Wo, a0, b0, l);
            bilin_src(wo0 + UT_W - 1, Wi, (float)Wi / (float)Wo, a1, b1, l);
            pw[i] = b1 - a0 + 1;
            q0[i] = a0;
            off[i] = used * UT_C;
            // stage [ph][pw][128 channels]
            const float* x = p.x[i] + (long)b * Hi * Wi * C + c0;
            const int items = ph * pw[i] * (UT_C / 4);
            for (int e = threadIdx.x; e < items; e += 256) {
                const int cq = e & (UT_C / 4 - 1), px = e >> 5, pr = px / pw[i], pc = px - pr * pw[i];
                *reinterpret_cast<float4*>(s_src + off[i] + px * UT_C + 4 * cq) =
                    *reinterpret_cast<const float4*>(x + ((long)(r0[i] + pr) * Wi + (q0[i] + pc)) * C + 4 * cq);
            }
            used += ph * pw[i];
        }
    }
    __syncthreads();
    const int cq = threadIdx.x & 31, colh = threadIdx.x >> 5;           // 32 channel quads x 8 columns per pass
#pragma unroll 2
    for (int it = 0; it < UT_H * 2; ++it) {
        const int ho = ho0 + (it >> 1), wo = wo0 + (it & 1) * 8 + colh;
        const long oi = ((((long)b * Ho + ho) * Wo) + wo) * C + c0 + 4 * cq;
        float4 o = base_ ? *reinterpret_cast<const float4*>(base_ + oi) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i >= p.n) break;
            const int Hi = p.Hi[i], Wi = p.Wi[i];
            int h0, h1, w0, w1; float lh, lw;
            bilin_src(ho, Hi, (float)Hi / (float)Ho, h0, h1, lh);
            bilin_src(wo, Wi, (float)Wi / (float)Wo, w0, w1, lw);
            const float c00 = (1.f - lh) * (1.f - lw), c01 = (1.f - lh) * lw, c10 = lh * (1.f - lw), c11 = lh * lw;
            const float* sp = s_src + off[i] + 4 * cq;
            const float4 v00 = *reinterpret_cast<const float4*>(sp + ((h0 - r0[i]) * pw[i] + (w0 - q0[i])) * UT_C), v01 = *reinterpret_cast<const float4*>(sp + ((h0 - r0[i]) * pw[i] + (w1 - q0[i])) * UT_C);
            const float4 v10 = *reinterpret_cast<const float4*>(sp + ((h1 - r0[i]) * pw[i] + (w0 - q0[i])) * UT_C), v11 = *reinterpret_cast<const float4*>(sp + ((h1 - r0[i]) * pw[i] + (w1 - q0[i])) * UT_C);
            float4 uu;
            uu.x = c00 * v00.x + c01 * v01.x + c10 * v10.x + c11 * v11.x;
            uu.y = c00 * v00.y + c01 * v01.y + c10 * v10.y + c11 * v11.y;
            uu.z = c00 * v00.z + c01 * v01.z + c10 * v10.z + c11 * v11.z;
            uu.w = c00 * v00.w + c01 * v01.w + c10 * v10.w + c11 * v11.w;
            o = (i == 0 && !base_) ? uu : f4_add(uu, o);
        }
        *reinterpret_cast<float4*>(y + oi) = o;
    }
}

// width folds of all sources: work item = (dy row o, source column j in [0, Wi_0 + Wi_1 + Wi_2), channel quad).  A dy row (Wo x C floats: 256 KB in the
// peer heads) is read by every output column it overlaps -- ~6 times in all -- so the workgroups of ONE row must share an L2: consecutive workgroup
// ids go round robin over the 8 XCDs, and a row's `bpr` workgroups were spread over all of them (PMC: 632 MB per launch for 134 MB of dy); the bijective
// remap below puts consecutive LOGICAL ids on one XCD.
__global__ __launch_bounds__(256) void upsample_multi_bwd_w_kernel(UpMulti p, const float* __restrict__ dy, long outer, int Wo, int C, int bpr) {
    const int QC = C >> 2;
    const int Wsum = p.Wi[0] + (p.n > 1 ? p.Wi[1] : 0) + (p.n > 2 ? p.Wi[2] : 0);
    const int nwg = (int)gridDim.x, qx = nwg >> 3, rx = nwg & 7, xcd = blockIdx.x & 7, ix = blockIdx.x >> 3;
    const int lb = (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + ix;
    const long o = lb / bpr;
    {
        const long e = (long)(lb % bpr) * blockDim.x + threadIdx.x;
        if (e >= (long)Wsum * QC) return;
        const int c = (int)(e % QC) * 4;
        int j = (int)(e / QC);
        int src = 0;
        if (j >= p.Wi[0]) { j -= p.Wi[0]; src = 1; if (j >= p.Wi[1]) { j -= p.Wi[1]; src = 2; } }
        const int Wi = p.Wi[src];
        const float sc = (float)Wi / (float)Wo;
        int lo, hi;
        out_range(j, Wi, Wo, lo, hi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* base = dy + o * Wo * C + c;
#pragma unroll 4
        for (int t = lo; t < hi; ++t) {
            int i0, i1; float l;
            bilin_src(t, Wi, sc, i0, i1, l);
            const float wgt = (i0 == j ? 1.f - l : 0.f) + (i1 == j ? l : 0.f);
            const float4 g = *reinterpret_cast<const float4*>(base + (long)t * C);
            acc.x = fmaf(wgt, g.x, acc.x); acc.y = fmaf(wgt, g.y, acc.y); acc.z = fmaf(wgt, g.z, acc.z); acc.w = fmaf(wgt, g.w, acc.w);
        }
        float* dsts[3] = {p.d[0], p.d[1], p.d[2]};
        *reinterpret_cast<float4*>(dsts[src] + (o * Wi + j) * C + c) = acc;
    }
}

// The same width folds with the dy row read ONCE (round 5): workgroup = (dy row o, 32 channels); the row's [Wo][32] slice is staged in LDS (whole 128-byte lines), then
// every (source column, channel quad) item folds its 2 s taps from there -- same taps, same order, same fmaf as above: bit-identical.  The kernel above reads every dy row
// ~6 times through L2 (252 MB of traffic per launch for 134 MB of dy in the peer heads: 71 us, eight launches on the two sweeps' chains of a bs=4 step).
__global__ __launch_bounds__(256) void upsample_multi_bwd_w_lds_kernel(UpMulti p, const float* __restrict__ dy, int Wo, int C) {
    extern __shared__ __attribute__((aligned(16))) float s_row[];          // [Wo][32]
    const long o = blockIdx.x;
    const int c0 = blockIdx.y * 32;
    const float* src = dy + o * Wo * C + c0;
    for (int e = threadIdx.x; e < Wo * 8; e += 256) {
        const int t = e >> 3, q = e & 7;
        *reinterpret_cast<float4*>(s_row + t * 32 + 4 * q) = *reinterpret_cast<const float4*>(src + (long)t * C + 4 * q);
    }
    __syncthreads();
    const int Wsum = p.Wi[0] + (p.n > 1 ? p.Wi[1] : 0) + (p.n > 2 ? p.Wi[2] : 0);
    float* dsts[3] = {p.d[0], p.d[1], p.d[2]};
    for (int e = threadIdx.x; e < Wsum * 8; e += 256) {
        const int q = e & 7;
        int j = e >> 3, sidx = 0;
        if (j >= p.Wi[0]) { j -= p.Wi[0]; sidx = 1; if (j >= p.Wi[1]) { j -= p.Wi[1]; sidx = 2; } }
        const int Wi = p.Wi[sidx];
        const float sc = (float)Wi / (float)Wo;
        int lo, hi;
        out_range(j, Wi, Wo, lo, hi);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int t = lo; t < hi; ++t) {
            int i0, i1; float l;
            bilin_src(t, Wi, sc, i0, i1, l);
            const float wgt = (i0 == j ? 1.f - l : 0.f) + (i1 == j ? l : 0.f);
            const float4 g = *reinterpret_cast<const float4*>(s_row + t * 32 + 4 * q);
            acc.x = fmaf(wgt, g.x, acc.x); acc.y = fmaf(wgt, g.y, acc.y); acc.z = fmaf(wgt, g.z, acc.z); acc.w = fmaf(wgt, g.w, acc.w);
        }
        *reinterpret_cast<float4*>(dsts[sidx] + (o * Wi + j) * C + c0 + 4 * q) = acc;
    }
}

// y = x * dropmask / (1 - p): one hash per aligned float4 (mdvit_drop_scale4), n % 4 == 0
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, uint32_t k0, uint32_t k1,
                                                      const uint32_t* __restrict__ seed, uint32_t thresh, float inv_keep) {
    const uint32_t k0e = k0 ^ (seed ? seed[0] : 0u), k1e = k1 + (seed ? seed[1] : 0u);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(x + 4 * i);
        const float4 d = mdvit_drop_scale4(k0e, k1e, (uint32_t)(4 * i), thresh, inv_keep);
        *reinterpret_cast<float4*>(y + 4 * i) = make_float4(v.x * d.x, v.y * d.y, v.z * d.z, v.w * d.w);
    }
}

inline int ew_grid(long total) { return (int)min((total + 255) / 256, 16384L); }

}  // namespace

extern "C" int mdvit_dwconv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Hi, int32_t Wi,
                                   int32_t C, int32_t stride, int32_t add_input, void* stream) {
    MDVIT_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 4 == 0 && C <= 4096, MDVIT_E_SHAPE, "dwconv3x3_fwd: bad shape B=%d H=%d W=%d C=%d", B, Hi, Wi, C);
    MDVIT_CHECK_ARG(stride == 1 || stride == 2, MDVIT_E_SHAPE, "dwconv3x3_fwd: stride must be 1 or 2");
    MDVIT_CHECK_ARG(!(add_input && stride != 1), MDVIT_E_SHAPE, "dwconv3x3_fwd: add_input needs stride 1");
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    if (stride == 1) {          // LDS-tiled path shared with the attention's window convolutions
        launch_conv_tile<3, false>(x, (long)C, 0, w, bias, y, (long)C, 0, CtGeom{B, Hi, Wi}, C, (hipStream_t)stream, add_input);
        MDVIT_LAUNCH_CHECK();
        return MDVIT_OK;
    }
    const long total = (long)B * Ho * Wo * C / 4;
    hipLaunchKernelGGL(dwconv3x3_fwd_kernel, dim3(ew_grid(total)), dim3(256), sizeof(float) * 9 * C, (hipStream_t)stream,
                       x, w, bias, y, B, Hi, Wi, C, stride, add_input);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_dwconv3x3_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias, void* ws, size_t ws_bytes,
                                   int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t stride, int32_t add_input, int32_t accumulate,
                                   void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 4 == 0 && C <= 1024, MDVIT_E_SHAPE, "dwconv3x3_bwd: bad shape B=%d H=%d W=%d C=%d", B, Hi, Wi, C);
    MDVIT_CHECK_ARG(stride == 1 || stride == 2, MDVIT_E_SHAPE, "dwconv3x3_bwd: stride must be 1 or 2");
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    if (stride == 1) {
        const CtGeom cg{B, Hi, Wi};
        if (dx) launch_conv_tile<3, true>(dy, (long)C, 0, w, nullptr, dx, (long)C, 0, cg, C, s, add_input);
        if (dw) {
            int tpb; long nblk;
            conv_wgrad_plan(cg, C, tpb, nblk);
            MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nblk, CT_CL * 10, "dwconv3x3_bwd");
            const int rc = launch_conv_tile_wgrad<3>(dy, (long)C, 0, x, (long)C, 0, dw, dbias, (float*)ws, cg, C, s, accumulate);
            if (rc != MDVIT_OK) return rc;
        }
        MDVIT_LAUNCH_CHECK();
        return MDVIT_OK;
    }
    if (dx) {
        const long total = (long)B * Hi * Wi * C / 4;
        hipLaunchKernelGGL(dwconv3x3_dgrad_kernel, dim3(ew_grid(total)), dim3(256), sizeof(float) * 9 * C, s, dy, w, dx, B, Hi, Wi, C, stride, add_input);
    }
    if (dw) {
        const long ntok = (long)B * Ho * Wo;
        int tpb = (int)max(64L, (ntok + 1023) / 1024);
        const int nblk = cdiv(ntok, tpb);
        MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nblk, 10 * C, "dwconv3x3_bwd");
        hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3(nblk), dim3(256), sizeof(float) * 10 * C, s, dy, x, (float*)ws, B, Hi, Wi, C, stride, tpb);
        const int rc = mdvit_reduce_partials((const float*)ws, nblk, 10L * C, 9 * C, dw, C, dbias, accumulate, s);
        if (rc != MDVIT_OK) return rc;
    }
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_gconv2_3x3_fwd(const float* skip, const float* up, const float* w, float* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    MDVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 2048, MDVIT_E_SHAPE, "gconv2_fwd: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
    const long total = (long)B * H * W * C / 2;
    static const bool tiled = [] { const char* e = getenv("MDVIT_GCONV2_TILES"); return !(e && e[0] == '0'); }();          // 0: the element-per-thread kernels (A/B)
    const int tiles_w = cdiv(W, CT_TW), tiles = tiles_w * cdiv(H, CT_TH);
    if (tiled && C % 32 == 0 && aligned16(skip) && aligned16(up) && B <= 65535)
        hipLaunchKernelGGL((gconv2_tile_kernel<false>), dim3(tiles, 2 * C / CT_CL, B), dim3(256), 0, (hipStream_t)stream, skip, up, w, y, (float*)nullptr, H, W, C, tiles_w);
    else
        hipLaunchKernelGGL(gconv2_fwd_kernel, dim3(ew_grid(total)), dim3(256), sizeof(float) * 18 * C, (hipStream_t)stream, skip, up, w, y, B, H, W, C);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_gconv2_3x3_bwd(const float* dy, const float* skip, const float* up, const float* w, float* dskip, float* dup, float* dw,
                                    void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t accumulate, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C <= 1024, MDVIT_E_SHAPE, "gconv2_bwd: bad shape B=%d H=%d W=%d C=%d", B, H, W, C);
    const long total = (long)B * H * W * C / 2;
    MDVIT_CHECK_ARG((dskip == nullptr) == (dup == nullptr), MDVIT_E_SHAPE, "gconv2_bwd: dskip and dup go together");
    static const bool tiled = [] { const char* e = getenv("MDVIT_GCONV2_TILES"); return !(e && e[0] == '0'); }();
    const int tiles_w = cdiv(W, CT_TW), tiles = tiles_w * cdiv(H, CT_TH);
    if (dskip && tiled && C % 64 == 0 && aligned16(dy) && B <= 65535)          // (C % 64: a 32-channel concat block reads 16 dy channels = whole float4 quads of one source half)
        hipLaunchKernelGGL((gconv2_tile_kernel<true>), dim3(tiles, 2 * C / CT_CL, B), dim3(256), 0, s, dy, (const float*)nullptr, w, dskip, dup, H, W, C, tiles_w);
    else if (dskip) hipLaunchKernelGGL(gconv2_dgrad_kernel, dim3(ew_grid(total)), dim3(256), sizeof(float) * 18 * C, s, dy, w, dskip, dup, B, H, W, C);
    if (dw) {
        // dw[g][j][tap] = sum dy[.., g] cat[.. + tap, 2g + j]: per concat half a depthwise-style weight gradient on the LDS tiles of
        // conv_tile.h (input read once instead of nine times), the gradient channel of input channel c being c / 2
        const CtGeom cg{B, H, W};
        int tpb; long nblk;
        conv_wgrad_plan(cg, C, tpb, nblk);
        if (C % 8 == 0 && ws != nullptr && ws_bytes >= sizeof(float) * (size_t)nblk * CT_CL * 10) {
            int rc = launch_conv_tile_wgrad<3, 2>(dy, (long)C, 0, skip, (long)C, 0, dw, nullptr, (float*)ws, cg, C, s, accumulate);
            if (rc == MDVIT_OK) rc = launch_conv_tile_wgrad<3, 2>(dy, (long)C, C / 2, up, (long)C, 0, dw + 9L * C, nullptr, (float*)ws, cg, C, s, accumulate);
            if (rc != MDVIT_OK) return rc;
        } else {
            const long ntok = (long)B * H * W;
            int tpb2 = (int)max(64L, (ntok + 1023) / 1024);
            const int nb = cdiv(ntok, tpb2);
            MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nb, 18 * C, "gconv2_bwd");
            hipLaunchKernelGGL(gconv2_wgrad_kernel, dim3(nb), dim3(256), sizeof(float) * 18 * C, s, dy, skip, up, (float*)ws, B, H, W, C, tpb2);
            const int rc = mdvit_reduce_partials((const float*)ws, nb, 18L * C, 18 * C, dw, 0, nullptr, accumulate, s);
            if (rc != MDVIT_OK) return rc;
        }
    }
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_im2col3x3(const float* x, float* col, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, int32_t stride, int32_t dilation, void* stream) {
    MDVIT_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Cin > 0 && (stride == 1 || stride == 2) && dilation >= 1 && (dilation == 1 || stride == 1), MDVIT_E_SHAPE,
                    "im2col3x3: bad shape (stride %d, dilation %d)", stride, dilation);
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(ew_grid((long)B * Ho * Wo * Cin)), dim3(256), 0, (hipStream_t)stream, x, col, B, Hi, Wi, Cin, stride, dilation);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_col2im3x3(const float* dcol, float* dx, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, int32_t stride, int32_t dilation, void* stream) {
    MDVIT_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Cin > 0 && (stride == 1 || stride == 2) && dilation >= 1 && (dilation == 1 || stride == 1), MDVIT_E_SHAPE,
                    "col2im3x3: bad shape (stride %d, dilation %d)", stride, dilation);
    hipLaunchKernelGGL(col2im3x3_kernel, dim3(ew_grid((long)B * Hi * Wi * Cin)), dim3(256), 0, (hipStream_t)stream, dcol, dx, B, Hi, Wi, Cin, stride, dilation);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_dropout_f32(const float* x, float* y, int64_t n, float p, uint32_t key0, uint32_t key1, const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(n > 0 && n % 4 == 0 && n < (1L << 32) && p >= 0.f && p < 1.f, MDVIT_E_SHAPE, "dropout: need 0 < n < 2^32, n %% 4 == 0, 0 <= p < 1 (n=%ld p=%g)", (long)n, p);
    MDVIT_CHECK_ARG(aligned16(x) && aligned16(y), MDVIT_E_ALIGN, "dropout: 16-byte aligned buffers");
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, (long)(n / 4), key0, key1, drop_seed,
                       mdvit_drop_thresh(p), 1.f / (1.f - p));
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_stemconv_fwd(const float* img, const float* w, float* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream) {
    MDVIT_CHECK_ARG(Cin == 3, MDVIT_E_SHAPE, "stemconv_fwd: only in_chans == 3 is built (got %d)", Cin);
    MDVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 4 == 0 && Cout <= 256, MDVIT_E_SHAPE, "stemconv_fwd: bad shape");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    static const bool px32 = [] { const char* e = getenv("MDVIT_STEM_FWD32"); return !(e && e[0] == '0'); }();          // 0: the channel-quad kernel (A/B)
    if (px32 && Cout == 32)
        hipLaunchKernelGGL((stemconv_fwd32_kernel<3>), dim3((int)min(((long)B * Ho * Wo + 127) / 128, 8192L)), dim3(256), 0, (hipStream_t)stream, img, w, y, B, H, W);
    else
        hipLaunchKernelGGL((stemconv_fwd_kernel<3>), dim3(ew_grid((long)B * Ho * Wo * Cout / 4)), dim3(256), sizeof(float) * 27 * Cout, (hipStream_t)stream,
                           img, w, y, B, H, W, Cout);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_stemconv_wgrad(const float* img, const float* dy, float* dw, void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W,
                                    int32_t Cin, int32_t Cout, int32_t accumulate, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(Cin == 3, MDVIT_E_SHAPE, "stemconv_wgrad: only in_chans == 3 is built (got %d)", Cin);
    MDVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout <= 256, MDVIT_E_SHAPE, "stemconv_wgrad: bad shape");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long npix = (long)B * Ho * Wo;
    int ppb = (int)max(64L, (npix + 2047) / 2048);
    const int nblk = cdiv(npix, ppb);
    MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nblk, 27 * Cout, "stemconv_wgrad");
    static const bool mfma = [] { const char* e = getenv("MDVIT_STEM_WGRAD_MFMA"); return !(e && e[0] == '0'); }();          // 0: the scalar kernel (A/B)
    if (mfma && Cout % 32 == 0) {                    // half the partial rows: a workgroup's pixels are walked by four waves with eight pairs in flight each
        ppb = (int)max(128L, (npix + 1023) / 1024);
        const int nb2 = cdiv(npix, ppb);
        hipLaunchKernelGGL((stemconv_wgrad_mfma_kernel<3>), dim3(nb2), dim3(256), 0, s, img, dy, (float*)ws, B, H, W, Cout, ppb);
        MDVIT_LAUNCH_CHECK();
        return mdvit_reduce_partials((const float*)ws, nb2, 27L * Cout, 27 * Cout, dw, 0, nullptr, accumulate, s);
    }
    hipLaunchKernelGGL((stemconv_wgrad_kernel<3>), dim3(nblk), dim3(256), sizeof(float) * 27 * Cout, s, img, dy, (float*)ws, B, H, W, Cout, ppb);
    MDVIT_LAUNCH_CHECK();
    return mdvit_reduce_partials((const float*)ws, nblk, 27L * Cout, 27 * Cout, dw, 0, nullptr, accumulate, s);
}

extern "C" int mdvit_upsample_fwd(const float* x, const float* base, float* y, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C, void* stream) {
    MDVIT_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, MDVIT_E_SHAPE, "upsample_fwd: bad shape");
    const long total = (long)B * Ho * Wo * ((C & 3) == 0 ? C / 4 : C);
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, base, y, B, Hi, Wi, Ho, Wo, C);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" size_t mdvit_upsample_bwd_ws_bytes(int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C) {
    if (B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0) return 0;
    return sizeof(float) * (size_t)B * Ho * Wi * C;          // tmp [B,Ho,Wi,C] between the two passes
}

extern "C" int mdvit_upsample_bwd(const float* dy, float* dx, void* ws, size_t ws_bytes, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                  int32_t C, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, MDVIT_E_SHAPE, "upsample_bwd: bad shape");
    MDVIT_CHECK_ARG(ws != nullptr && ws_bytes >= mdvit_upsample_bwd_ws_bytes(B, Hi, Wi, Ho, Wo, C), MDVIT_E_WORKSPACE,
                    "upsample_bwd: workspace too small (mdvit_upsample_bwd_ws_bytes)");
    float* tmp = (float*)ws;
    const int vec = (C & 3) == 0 && aligned16(dy) && aligned16(dx) && aligned16(tmp);
    const long cq = vec ? C / 4 : C;
    // pass W: dy [B*Ho][Wo][C] -> tmp [B*Ho][Wi][C]
    {
        const long total = (long)B * Ho * Wi * cq;
        hipLaunchKernelGGL((upsample_bwd_pass_kernel<true>), dim3(ew_grid(total)), dim3(256), 0, s, dy, tmp, (long)B * Ho, Wi, Wo, cq, vec);
    }
    // pass H: tmp [B][Ho][Wi*C] -> dx [B][Hi][Wi*C]
    {
        const long total = (long)B * Hi * Wi * cq;
        hipLaunchKernelGGL((upsample_bwd_pass_kernel<false>), dim3(ew_grid(total)), dim3(256), 0, s, tmp, dx, (long)B, Hi, Ho, (long)Wi * cq, vec);
    }
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

/* see include/mdvit_hip.h */
extern "C" int mdvit_upsample_multi_fwd(const float* const* xs, const int32_t* Hi, const int32_t* Wi, int32_t n, const float* base, float* y, int32_t B, int32_t Ho,
                                        int32_t Wo, int32_t C, void* stream) {
    MDVIT_CHECK_ARG(n >= 1 && n <= 3 && B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, MDVIT_E_SHAPE, "upsample_multi_fwd: 1..3 sources, C %% 4 == 0 (n=%d C=%d)", n, C);
    UpMulti p; memset(&p, 0, sizeof(p));
    p.n = n;
    for (int i = 0; i < n; ++i) {
        MDVIT_CHECK_ARG(xs[i] && Hi[i] > 0 && Wi[i] > 0 && aligned16(xs[i]), MDVIT_E_SHAPE, "upsample_multi_fwd: bad source %d", i);
        p.x[i] = xs[i]; p.Hi[i] = Hi[i]; p.Wi[i] = Wi[i];
    }
    MDVIT_CHECK_ARG(aligned16(y) && (!base || aligned16(base)), MDVIT_E_ALIGN, "upsample_multi_fwd: y / base must be 16-byte aligned");
    {   // the LDS-tiled kernel where the shape allows it (the peer heads: 128 x 128 x 512 from 64 / 32 / 16)
        static const bool tile_on = !(getenv("MDVIT_UPSAMPLE_TILED") && atoi(getenv("MDVIT_UPSAMPLE_TILED")) == 0);
        bool ok = tile_on && Ho % UT_H == 0 && Wo % UT_W == 0 && C % UT_C == 0 && (long)(Ho / UT_H) * (Wo / UT_W) < (1L << 30) && C / UT_C < 65536 && B < 65536;
        int pool = 0;
        for (int i = 0; i < n && ok; ++i) {
            // power-of-two factors only: for them a tile's source patch is at most tile / factor + 2 rows (columns) -- the bound the pool is sized with; a factor of 3
            // puts one more row under some tiles (105 staged pixels against 96 estimated), and nothing in the models asks for it
            const int fh = Ho / Hi[i], fw = Wo / Wi[i];
            ok = Ho % Hi[i] == 0 && Wo % Wi[i] == 0 && fh >= 2 && fw >= 2 && fh <= 16 && fw <= 16 && (fh & (fh - 1)) == 0 && (fw & (fw - 1)) == 0;
            if (ok) pool += (UT_H / fh + 2) * (UT_W / fw + 2);         // rows h0(first) .. h1(last) of the tile: at most tile / factor + 2
        }
        if (ok && pool <= UT_POOL) {
            hipLaunchKernelGGL(upsample_multi_tile_kernel, dim3((Ho / UT_H) * (Wo / UT_W), C / UT_C, B), dim3(256), 0, (hipStream_t)stream, p, base, y, Ho, Wo, C);
            MDVIT_LAUNCH_CHECK();
            return MDVIT_OK;
        }
    }
    MDVIT_CHECK_ARG((long)Wo * (C / 4) < (1L << 30), MDVIT_E_SHAPE, "upsample_multi_fwd: output row too long");
    const long nunits = (long)B * Ho * (((long)Wo * (C / 4) + 1023) >> 10);
    hipLaunchKernelGGL(upsample_multi_fwd_kernel, dim3((unsigned)min(nunits, 16384L)), dim3(256), 0, (hipStream_t)stream, p, base, y, B, Ho, Wo, C);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" size_t mdvit_upsample_multi_bwd_ws_bytes(const int32_t* Wi, int32_t n, int32_t B, int32_t Ho, int32_t C) {
    size_t t = 0;
    for (int i = 0; i < n && i < 3; ++i) t += (size_t)B * Ho * Wi[i] * C;
    return sizeof(float) * t;
}

extern "C" int mdvit_upsample_multi_bwd(const float* dy, float* const* dxs, const int32_t* Hi, const int32_t* Wi, int32_t n, void* ws, size_t ws_bytes, int32_t B,
                                        int32_t Ho, int32_t Wo, int32_t C, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(n >= 1 && n <= 3 && B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, MDVIT_E_SHAPE, "upsample_multi_bwd: 1..3 sources, C %% 4 == 0 (n=%d C=%d)", n, C);
    MDVIT_CHECK_ARG(ws && ws_bytes >= mdvit_upsample_multi_bwd_ws_bytes(Wi, n, B, Ho, C) && aligned16(ws) && aligned16(dy), MDVIT_E_WORKSPACE,
                    "upsample_multi_bwd: workspace too small (mdvit_upsample_multi_bwd_ws_bytes)");
    UpMulti p; memset(&p, 0, sizeof(p));
    p.n = n;
    float* t = (float*)ws;
    int wsum = 0;
    for (int i = 0; i < n; ++i) {
        MDVIT_CHECK_ARG(dxs[i] && Hi[i] > 0 && Wi[i] > 0 && aligned16(dxs[i]), MDVIT_E_SHAPE, "upsample_multi_bwd: bad source %d", i);
        p.Hi[i] = Hi[i]; p.Wi[i] = Wi[i]; p.d[i] = t;
        t += (size_t)B * Ho * Wi[i] * C;
        wsum += Wi[i];
    }
    const long cq = C / 4;
    const int bpr = (int)cdiv((long)wsum * cq, 256L);
    MDVIT_CHECK_ARG((long)B * Ho * bpr < (1L << 31), MDVIT_E_SHAPE, "upsample_multi_bwd: too many workgroups");
    static const bool lds_rows = [] { const char* e = getenv("MDVIT_UPSAMPLE_BWD_LDS"); return !(e && e[0] == '0'); }();          // 0: the L2 re-reading kernel (A/B)
    if (lds_rows && C % 32 == 0 && Wo <= 384 && (long)B * Ho < (1L << 31) && C / 32 <= 65535)
        hipLaunchKernelGGL(upsample_multi_bwd_w_lds_kernel, dim3((unsigned)((long)B * Ho), C / 32), dim3(256), sizeof(float) * Wo * 32, s, p, dy, Wo, C);
    else
        hipLaunchKernelGGL(upsample_multi_bwd_w_kernel, dim3((unsigned)((long)B * Ho * bpr)), dim3(256), 0, s, p, dy, (long)B * Ho, Wo, C, bpr);
    for (int i = 0; i < n; ++i)          // pass H per source: tmp_i [B][Ho][Wi*C] -> dx_i [B][Hi][Wi*C]
        hipLaunchKernelGGL((upsample_bwd_pass_kernel<false>), dim3(ew_grid((long)B * Hi[i] * Wi[i] * cq)), dim3(256), 0, s, p.d[i], dxs[i], (long)B, Hi[i], Ho,
                           (long)Wi[i] * cq, 1);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
