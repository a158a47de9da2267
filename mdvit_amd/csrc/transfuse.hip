// Kernels of the TransFuse_S_adapt path (BASELINE configs[4]; reference: Models/Hybrid_models/TransFuseFolder/TransFuse.py,
// vision_transformer.py, DeiT.py, multi_train_TransFuse.py) that the MDViT path does not already provide.  Activations are NHWC
// fp32 as everywhere in this library.  Dense 3x3 / 1x1 convolutions, BatchNorm (+ReLU), LayerNorm, the Linear / MLP GEMMs and the
// Domain Adapter reuse the MDViT kernels; what is new here:
//   image stem conv KxK stride 2 (ResNet conv1 7x7, torchvision resnet)           max-pool 3x3 / 2 (ResNet)
//   softmax(Q K^T) V with the head-softmax Domain Adapter (Attention_Sup, vision_transformer.py:148-169)
//   bilinear resize with align_corners=True (Up, the three heads: TransFuse.py:528,267-269)
//   ChannelPool + 7x7 (2->1) spatial attention, sigmoid gates, element-wise add+ReLU / product (BiFusion_block, Attention_block,
//   DoubleConv, BasicBlock), single-channel BatchNorm, stride-2 pixel pick (1x1 stride-2 shortcut convs), patch gather (PatchEmbed),
//   positional-embedding add, Dropout2d, and structure_loss (31x31 box filter + weighted BCE / IoU, multi_train_TransFuse.py:29-38).
// These are plain grid-stride kernels: correctness first -- this path is measured (bench.py --model transfuse), not yet tuned.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
inline int tf_grid(long n, int per_block = 256) { long g = (n + per_block - 1) / per_block; return (int)(g < 1 ? 1 : (g > 65535 * 4 ? 65535 * 4 : g)); }

// ---- image conv KSxKS stride 2 pad KS/2: NCHW image -> NHWC; thread = (output pixel, 4 output channels) -------------------------
template <int CIN, int KS>
__global__ __launch_bounds__(256) void imgconv_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w, float* __restrict__ y,
                                                          int B, int H, int W, int Cout) {
    extern __shared__ float s_w[];   // [CIN*KS*KS][Cout]
    constexpr int KK = CIN * KS * KS, PAD = KS / 2;
    for (int i = threadIdx.x; i < Cout * KK; i += blockDim.x) { const int co = i / KK, k = i % KK; s_w[k * Cout + co] = w[i]; }
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
        for (int ci = 0; ci < CIN; ++ci)
            for (int kh = 0; kh < KS; ++kh) {
                const int hi = 2 * ho + kh - PAD;
                if (hi < 0 || hi >= H) continue;
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    const int wi = 2 * wo + kw - PAD;
                    if (wi < 0 || wi >= W) continue;
                    const float xv = img[(((long)b * CIN + ci) * H + hi) * W + wi];
                    const float4 wv = *reinterpret_cast<const float4*>(&s_w[((ci * KS + kh) * KS + kw) * Cout + co]);
                    acc.x = fmaf(xv, wv.x, acc.x); acc.y = fmaf(xv, wv.y, acc.y); acc.z = fmaf(xv, wv.z, acc.z); acc.w = fmaf(xv, wv.w, acc.w);
                }
            }
        *reinterpret_cast<float4*>(y + (((long)b * Ho + ho) * Wo + wo) * Cout + co) = acc;
    }
}

// dw[co][ci][kh][kw] = sum over output pixels of dy[b,ho,wo,co] * img[b,ci,2ho+kh-PAD,2wo+kw-PAD].  A workgroup owns a run of output
// pixels; thread t owns weight elements t, t+256, ... of the [Cout][KK] matrix and walks the run (dy and the image patch are L1 hits);
// one row of partial sums per workgroup, reduced in fixed order afterwards.
template <int CIN, int KS>
__global__ __launch_bounds__(256) void imgconv_wgrad_kernel(const float* __restrict__ img, const float* __restrict__ dy, float* __restrict__ part,
                                                            int B, int H, int W, int Cout, int pix_per_block) {
    constexpr int KK = CIN * KS * KS, PAD = KS / 2;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long npix = (long)B * Ho * Wo;
    const long p_beg = (long)blockIdx.x * pix_per_block, p_end = min(npix, p_beg + pix_per_block);
    const int nw = Cout * KK;
    for (int e0 = threadIdx.x; e0 < nw; e0 += blockDim.x) {
        const int co = e0 / KK, k = e0 % KK, ci = k / (KS * KS), kh = (k / KS) % KS, kw = k % KS;
        float acc = 0.f;
        for (long pix = p_beg; pix < p_end; ++pix) {
            const int wo = (int)(pix % Wo), ho = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
            const int hi = 2 * ho + kh - PAD, wi = 2 * wo + kw - PAD;
            if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
            acc = fmaf(dy[pix * Cout + co], img[(((long)b * CIN + ci) * H + hi) * W + wi], acc);
        }
        part[(long)blockIdx.x * nw + e0] = acc;
    }
}

// im2col of the same conv (rows = output pixels, columns (ci, kh, kw) padded to ldc with zeros): the stem conv and its weight gradient
// then run as GEMMs on the matrix cores (the direct kernels above: 0.3 ms forward, 3 ms weight gradient at 8 x 256 x 256)
template <int CIN, int KS>
__global__ __launch_bounds__(256) void imgconv_im2col_kernel(const float* __restrict__ img, float* __restrict__ col, int B, int H, int W, int ldc) {
    constexpr int KK = CIN * KS * KS, PAD = KS / 2;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo * ldc;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e % ldc);
        long r = e / ldc;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float v = 0.f;
        if (k < KK) {
            const int ci = k / (KS * KS), kh = (k / KS) % KS, kw = k % KS;
            const int hi = 2 * ho + kh - PAD, wi = 2 * wo + kw - PAD;
            if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = img[(((long)b * CIN + ci) * H + hi) * W + wi];
        }
        col[e] = v;
    }
}

// ---- max-pool 3x3 stride 2 pad 1 (NHWC); idx = winning tap 0..8 (first maximum in (kh, kw) order, as ATen) -----------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int B, int H, int W, int C) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float best = -INFINITY; int bi = 0;
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = 2 * ho + kh - 1;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = 2 * wo + kw - 1;
                if (wi < 0 || wi >= W) continue;
                const float v = x[(((long)b * H + hi) * W + wi) * C + c];
                if (v > best || v != v) { best = v; bi = kh * 3 + kw; }
            }
        }
        y[e] = best; idx[e] = (uint8_t)bi;
    }
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                          int B, int H, int W, int C) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * H * W * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        const int wi = (int)(r % W); r /= W;
        const int hi = (int)(r % H);
        const int b = (int)(r / H);
        float acc = 0.f;
        // output windows that contain (hi, wi): ho with 2ho-1 <= hi <= 2ho+1
        for (int ho = (hi + 1 - 2 + 1) / 2; ho <= (hi + 1) / 2; ++ho) {
            if (ho < 0 || ho >= Ho || hi - (2 * ho - 1) < 0 || hi - (2 * ho - 1) > 2) continue;
            for (int wo = (wi + 1 - 2 + 1) / 2; wo <= (wi + 1) / 2; ++wo) {
                if (wo < 0 || wo >= Wo || wi - (2 * wo - 1) < 0 || wi - (2 * wo - 1) > 2) continue;
                const long o = (((long)b * Ho + ho) * Wo + wo) * C + c;
                if (idx[o] == (hi - (2 * ho - 1)) * 3 + (wi - (2 * wo - 1))) acc += dy[o];
            }
        }
        dx[e] = acc;
    }
}

// ---- bilinear resize, align_corners=True: src = dst * (in-1)/(out-1) ------------------------------------------------------------
__device__ __forceinline__ void ac_coord(int o, int in, int out, int& i0, int& i1, float& f) {
    const float s = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float src = s * (float)o;
    i0 = (int)src; i0 = i0 < in - 1 ? i0 : in - 1;
    i1 = i0 + 1 < in ? i0 + 1 : in - 1;
    f = src - (float)i0;
}
__global__ __launch_bounds__(256) void resize_ac_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const long total = (long)B * Ho * Wo * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int b = (int)(r / Ho);
        int h0, h1, w0, w1; float fh, fw;
        ac_coord(ho, Hi, Ho, h0, h1, fh); ac_coord(wo, Wi, Wo, w0, w1, fw);
        const float* xb = x + (long)b * Hi * Wi * C + c;
        const float v00 = xb[((long)h0 * Wi + w0) * C], v01 = xb[((long)h0 * Wi + w1) * C], v10 = xb[((long)h1 * Wi + w0) * C], v11 = xb[((long)h1 * Wi + w1) * C];
        // ATen's upsample_bilinear2d: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (w0lambda * v10 + w1lambda * v11)
        y[e] = (1.f - fh) * ((1.f - fw) * v00 + fw * v01) + fh * ((1.f - fw) * v10 + fw * v11);
    }
}
// backward as a GATHER (deterministic, no atomics): input pixel (hi, wi) collects from every output pixel whose two taps include it
__global__ __launch_bounds__(256) void resize_ac_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const long total = (long)B * Hi * Wi * C;
    const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        long r = e / C;
        const int wi = (int)(r % Wi); r /= Wi;
        const int hi = (int)(r % Hi);
        const int b = (int)(r / Hi);
        // candidate output rows: src = ho * sh in (hi - 1, hi + 1)
        const int ho_lo = sh > 0.f ? max(0, (int)floorf((float)(hi - 1) / sh) - 1) : 0, ho_hi = sh > 0.f ? min(Ho - 1, (int)ceilf((float)(hi + 1) / sh) + 1) : Ho - 1;
        const int wo_lo = sw > 0.f ? max(0, (int)floorf((float)(wi - 1) / sw) - 1) : 0, wo_hi = sw > 0.f ? min(Wo - 1, (int)ceilf((float)(wi + 1) / sw) + 1) : Wo - 1;
        float acc = 0.f;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            int h0, h1; float fh;
            ac_coord(ho, Hi, Ho, h0, h1, fh);
            float wh = 0.f;
            if (h0 == hi) wh += 1.f - fh;
            if (h1 == hi) wh += fh;
            if (wh == 0.f) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                int w0, w1; float fw;
                ac_coord(wo, Wi, Wo, w0, w1, fw);
                float ww = 0.f;
                if (w0 == wi) ww += 1.f - fw;
                if (w1 == wi) ww += fw;
                if (ww != 0.f) acc = fmaf(dy[(((long)b * Ho + ho) * Wo + wo) * C + c], wh * ww, acc);
            }
        }
        dx[e] = acc;
    }
}

// ---- element-wise -----------------------------------------------------------------------------------------------------------------
// mode 0: y = relu(a + b) (b may be null)   1: y = a * b   2: y = g * (yref > 0)   3: y = a + b
__global__ __launch_bounds__(256) void ew_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n, int mode) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float av = a[e], bv = b ? b[e] : 0.f;
        float r;
        if (mode == 0) r = fmaxf(av + bv, 0.f);
        else if (mode == 1) r = av * bv;
        else if (mode == 2) r = bv > 0.f ? av : 0.f;
        else r = av + bv;
        y[e] = r;
    }
}
// y = (a + b) + c: the three gradients of a tensor with three consumers (every encoder stage's output: next stage, decoder skip, peer heads) in ONE pass
// (two passes of ew_kernel mode 3 before: one launch and a third of the bytes fewer; the same two additions in the same order)
__global__ __launch_bounds__(256) void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float* __restrict__ y, long n4, long n) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const float4 av = reinterpret_cast<const float4*>(a)[e], bv = reinterpret_cast<const float4*>(b)[e], cv = reinterpret_cast<const float4*>(c)[e];
        reinterpret_cast<float4*>(y)[e] = make_float4((av.x + bv.x) + cv.x, (av.y + bv.y) + cv.y, (av.z + bv.z) + cv.z, (av.w + bv.w) + cv.w);
    }
    for (long e = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) y[e] = (a[e] + b[e]) + c[e];
}
// y = (a + b) + [p_0 | p_1 | ...]: the same fan-in where the third gradient arrives as G equal row-range parts (the peer heads' gradients of one encoder feature): the
// concatenation of the parts is never written (b may be null: y = a + parts).  Static-index select of the part pointer (no scratch).
struct AddPartsArgs { const float* p[8]; };
__global__ __launch_bounds__(256) void add_parts_kernel(const float* __restrict__ a, const float* __restrict__ b, AddPartsArgs parts, float* __restrict__ y, long part4, long n4) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        const int g = (int)(e / part4);
        const long o = e - (long)g * part4;
        const float* pp = parts.p[0];
#pragma unroll
        for (int i = 1; i < 8; ++i)
            if (g == i) pp = parts.p[i];
        const float4 av = reinterpret_cast<const float4*>(a)[e], cv = reinterpret_cast<const float4*>(pp)[o];
        float4 r = av;
        if (b) { const float4 bv = reinterpret_cast<const float4*>(b)[e]; r = make_float4(av.x + bv.x, av.y + bv.y, av.z + bv.z, av.w + bv.w); }
        reinterpret_cast<float4*>(y)[e] = make_float4(r.x + cv.x, r.y + cv.y, r.z + cv.z, r.w + cv.w);
    }
}
// y[b, r] = x[b, r] + pe[r]   (positional embedding, broadcast over the batch)
__global__ __launch_bounds__(256) void add_bcast_kernel(const float* __restrict__ x, const float* __restrict__ pe, float* __restrict__ y, long R, long n) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) y[e] = x[e] + pe[e % R];
}
// out[r] = sum_b g[b, r]  (fixed order)
__global__ __launch_bounds__(256) void sum_batch_kernel(const float* __restrict__ g, float* __restrict__ out, int B, long R) {
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += g[(long)b * R + r];
        out[r] = s;
    }
}
// x [B, P, C]; mode 0: s [B, P] (spatial gate), mode 1: s [B, C] (channel gate):  y = sigmoid(s) * x
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, long P, int C, long n, int mode) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long pix = e / C; const int c = (int)(e % C);
        const float sv = mode == 0 ? s[pix] : s[(pix / P) * C + c];
        y[e] = sigmoidf_(sv) * x[e];
    }
}
// dx = g * sigmoid(s);  spatial: ds[b,p] = sig (1 - sig) * sum_c g x   (one wave per pixel)
__global__ __launch_bounds__(256) void gate_bwd_spatial_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ s,
                                                               float* __restrict__ dx, float* __restrict__ ds, long npix, int C) {
    const int lane = threadIdx.x & 63;
    for (long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (long)gridDim.x * 4) {
        const float sg = sigmoidf_(s[pix]);
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float gv = g[pix * C + c], xv = x[pix * C + c];
            dx[pix * C + c] = gv * sg;
            acc = fmaf(gv, xv, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) ds[pix] = acc * sg * (1.f - sg);
    }
}
// channel: ds[b,c] = sig (1 - sig) * sum_p g x.  Pass 1: block = (64 channels, pixel chunk, b) writes dx and ONE partial row per chunk;
// pass 2 adds the chunks in order and applies sigmoid'.
__global__ __launch_bounds__(256) void gate_bwd_channel_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ s,
                                                               float* __restrict__ dx, float* __restrict__ part, long P, int C, int chunk) {
    __shared__ float red[4][64];
    const int b = blockIdx.z, c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
    const long p0 = (long)blockIdx.y * chunk, p1 = min(P, p0 + chunk);
    float acc = 0.f;
    if (c < C) {
        const float sg = sigmoidf_(s[(long)b * C + c]);
        for (long p = p0 + pl; p < p1; p += 4) {
            const long o = ((long)b * P + p) * C + c;
            const float gv = g[o];
            dx[o] = gv * sg;
            acc = fmaf(gv, x[o], acc);
        }
    }
    red[pl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (pl == 0 && c < C)
        part[((long)b * gridDim.y + blockIdx.y) * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void gate_bwd_channel_final_kernel(const float* __restrict__ part, const float* __restrict__ s, float* __restrict__ ds, int B, int C, int nchunk) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i % C;
    float acc = 0.f;
    for (int k = 0; k < nchunk; ++k) acc += part[((long)b * nchunk + k) * C + c];
    const float sg = sigmoidf_(s[i]);
    ds[i] = acc * sg * (1.f - sg);
}
// ChannelPool (TransFuse.py:20-22): y[m] = (max_c x, mean_c x); idx = first argmax.  One wave per pixel.
__global__ __launch_bounds__(256) void chanpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ idx, long M, int C) {
    const int lane = threadIdx.x & 63;
    for (long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long)gridDim.x * 4) {
        float best = -INFINITY, sum = 0.f; int bi = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = x[m * C + c];
            sum += v;
            if (v > best) { best = v; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        sum = wave_sum(sum);
        if (lane == 0) { y[2 * m] = best; y[2 * m + 1] = sum / (float)C; idx[m] = bi; }
    }
}
__global__ __launch_bounds__(256) void chanpool_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx, float* __restrict__ dx, long M, int C) {
    const long n = M * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long m = e / C; const int c = (int)(e % C);
        dx[e] = dy[2 * m + 1] / (float)C + (c == idx[m] ? dy[2 * m] : 0.f);
    }
}
// 7x7 conv, 2 -> 1 channels, pad 3, no bias (BiFusion_block.spatial, TransFuse.py:37): x [B,H,W,2], w [1,2,7,7], y [B,H,W]
__global__ __launch_bounds__(256) void conv7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int B, int H, int W) {
    __shared__ float sw[98];
    if (threadIdx.x < 98) sw[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const long total = (long)B * H * W;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int wo = (int)(e % W), ho = (int)((e / W) % H), b = (int)(e / ((long)W * H));
        float acc = 0.f;
        for (int kh = 0; kh < 7; ++kh) {
            const int hi = ho + kh - 3;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < 7; ++kw) {
                const int wi = wo + kw - 3;
                if (wi < 0 || wi >= W) continue;
                const float2 v = *reinterpret_cast<const float2*>(x + (((long)b * H + hi) * W + wi) * 2);
                acc = fmaf(v.x, sw[kh * 7 + kw], acc); acc = fmaf(v.y, sw[49 + kh * 7 + kw], acc);
            }
        }
        y[e] = acc;
    }
}
__global__ __launch_bounds__(256) void conv7_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int B, int H, int W) {
    __shared__ float sw[98];
    if (threadIdx.x < 98) sw[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const long total = (long)B * H * W;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int wi = (int)(e % W), hi = (int)((e / W) % H), b = (int)(e / ((long)W * H));
        float a0 = 0.f, a1 = 0.f;
        for (int kh = 0; kh < 7; ++kh) {
            const int ho = hi - kh + 3;
            if (ho < 0 || ho >= H) continue;
            for (int kw = 0; kw < 7; ++kw) {
                const int wo = wi - kw + 3;
                if (wo < 0 || wo >= W) continue;
                const float g = dy[((long)b * H + ho) * W + wo];
                a0 = fmaf(g, sw[kh * 7 + kw], a0); a1 = fmaf(g, sw[49 + kh * 7 + kw], a1);
            }
        }
        *reinterpret_cast<float2*>(dx + e * 2) = make_float2(a0, a1);
    }
}
// dw[c][kh][kw] = sum dy[b,ho,wo] x[b,ho+kh-3,wo+kw-3,c]: one workgroup per weight element (98), fixed-order block reduction
__global__ __launch_bounds__(256) void conv7_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, int B, int H, int W) {
    __shared__ double red[256];
    const int k = blockIdx.x, c = k / 49, kh = (k % 49) / 7, kw = k % 7;
    const long total = (long)B * H * W;
    double acc = 0.0;
    for (long e = threadIdx.x; e < total; e += 256) {
        const int wo = (int)(e % W), ho = (int)((e / W) % H), b = (int)(e / ((long)W * H));
        const int hi = ho + kh - 3, wi = wo + kw - 3;
        if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
        acc += (double)dy[e] * (double)x[(((long)b * H + hi) * W + wi) * 2 + c];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) dw[k] = (float)red[0];
}

// ---- BatchNorm over ONE channel (spatial.bn, psi.1): a single workgroup; statistics in double.  `groups` equal consecutive slices of
// the batch are normalised separately, in order (the domain-batched forward: each domain batch keeps its own statistics and the running
// statistics are updated once per domain batch, as four separate forwards would)
__global__ __launch_bounds__(1024) void bn1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ rm, float* __restrict__ rv, long long* __restrict__ nbt, float* __restrict__ y,
                                                       float* __restrict__ stat /* [groups][mean, rstd] */, long M, int groups, int training, float eps, float momentum) {
    __shared__ double r1[1024], r2[1024];
    __shared__ float s_mean, s_rstd;
    const long Mg = M / groups;
    const float ga = gamma[0], be = beta[0];
    for (int gidx = 0; gidx < groups; ++gidx) {
        const float* xg = x + gidx * Mg;
        float* yg = y + gidx * Mg;
        __syncthreads();
        if (training) {
            double a = 0.0, b = 0.0;
            for (long e = threadIdx.x; e < Mg; e += 1024) { const double v = xg[e]; a += v; b += v * v; }
            r1[threadIdx.x] = a; r2[threadIdx.x] = b;
            __syncthreads();
            for (int o = 512; o > 0; o >>= 1) { if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; } __syncthreads(); }
            if (threadIdx.x == 0) {
                const double mean = r1[0] / (double)Mg;
                double var = r2[0] / (double)Mg - mean * mean; var = var > 0.0 ? var : 0.0;
                s_mean = (float)mean; s_rstd = (float)(1.0 / sqrt(var + (double)eps));
                const double unbiased = Mg > 1 ? var * (double)Mg / (double)(Mg - 1) : var;
                rm[0] = (1.f - momentum) * rm[0] + momentum * (float)mean;
                rv[0] = (1.f - momentum) * rv[0] + momentum * (float)unbiased;
                if (nbt) nbt[0] += 1;
            }
        } else if (threadIdx.x == 0) {
            s_mean = rm[0]; s_rstd = 1.f / sqrtf(rv[0] + eps);
        }
        __syncthreads();
        const float mean = s_mean, rstd = s_rstd;
        if (threadIdx.x == 0) { stat[2 * gidx] = mean; stat[2 * gidx + 1] = rstd; }
        for (long e = threadIdx.x; e < Mg; e += 1024) yg[e] = (xg[e] - mean) * rstd * ga + be;
    }
}
__global__ __launch_bounds__(1024) void bn1_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ stat, float* __restrict__ dx, float* __restrict__ dgb /* dgamma, dbeta */,
                                                       long M, int groups, int training) {
    __shared__ double r1[1024], r2[1024];
    const long Mg = M / groups;
    const float ga = gamma[0];
    double dgam = 0.0, dbet = 0.0;
    for (int gidx = 0; gidx < groups; ++gidx) {
        const float mean = stat[2 * gidx], rstd = stat[2 * gidx + 1];
        const float* gg = g + gidx * Mg; const float* xg = x + gidx * Mg; float* dxg = dx + gidx * Mg;
        double a = 0.0, b = 0.0;
        for (long e = threadIdx.x; e < Mg; e += 1024) { const double gv = gg[e]; a += gv; b += gv * (double)((xg[e] - mean) * rstd); }
        __syncthreads();
        r1[threadIdx.x] = a; r2[threadIdx.x] = b;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) { if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; } __syncthreads(); }
        const float sg = (float)r1[0], sgx = (float)r2[0];
        dgam += r2[0]; dbet += r1[0];
        const float invM = 1.f / (float)Mg;
        for (long e = threadIdx.x; e < Mg; e += 1024) {
            const float xh = (xg[e] - mean) * rstd;
            dxg[e] = training ? ga * rstd * (gg[e] - sg * invM - xh * sgx * invM) : ga * rstd * gg[e];
        }
    }
    if (threadIdx.x == 0) { dgb[0] = (float)dgam; dgb[1] = (float)dbet; }
}

// ---- stride-2 pixel pick (the 1x1 stride-2 shortcut convolutions of ResNet) ------------------------------------------------------
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C, int backward) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (!backward) {
        const long total = (long)B * Ho * Wo * C;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
            const int c = (int)(e % C); long r = e / C;
            const int wo = (int)(r % Wo); r /= Wo; const int ho = (int)(r % Ho); const int b = (int)(r / Ho);
            dst[e] = src[(((long)b * H + 2 * ho) * W + 2 * wo) * C + c];
        }
    } else {        // dst [B,H,W,C] <- src [B,Ho,Wo,C] scattered, zeros elsewhere
        const long total = (long)B * H * W * C;
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
            const int c = (int)(e % C); long r = e / C;
            const int wi = (int)(r % W); r /= W; const int hi = (int)(r % H); const int b = (int)(r / H);
            dst[e] = ((hi | wi) & 1) ? 0.f : src[(((long)b * Ho + hi / 2) * Wo + wi / 2) * C + c];
        }
    }
}
// PatchEmbed gather (vision_transformer.py:233-240): NCHW image -> [B * (H/p) * (W/p), C*p*p] rows in (c, ky, kx) order
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int Cin, int H, int W, int p) {
    const int Hp = H / p, Wp = W / p, K = Cin * p * p;
    const long total = (long)B * Hp * Wp * K;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e % K); long r = e / K;
        const int px = (int)(r % Wp); r /= Wp; const int py = (int)(r % Hp); const int b = (int)(r / Hp);
        const int c = k / (p * p), ky = (k / p) % p, kx = k % p;
        out[e] = img[(((long)b * Cin + c) * H + py * p + ky) * W + px * p + kx];
    }
}
// Dropout2d: one keep decision per (sample, channel); x [B, P, C]
__global__ __launch_bounds__(256) void dropout2d_kernel(const float* __restrict__ x, float* __restrict__ y, long P, int C, long n, uint32_t k0, uint32_t k1,
                                                        uint32_t thresh, float inv_keep, const uint32_t* __restrict__ seed) {
    uint32_t s0 = 0, s1 = 0;
    if (seed) { s0 = seed[0]; s1 = seed[1]; }
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long b = e / (P * C); const int c = (int)(e % C);
        y[e] = x[e] * mdvit_drop_scale(k0 ^ s0, k1 + s1, (uint32_t)(b * C + c), thresh, inv_keep);
    }
}

// ---- softmax(Q K^T * scale) V, times the Domain Adapter's per-(head, channel) scale (vision_transformer.py:148-169) -------------
// qkv [B, N, 3C] as the qkv Linear writes it (q | k | v, channel = head * D + d).  P [B, H, N, N] is kept for the backward.
// forward: one workgroup = (b, head, 32 query rows); K, then V, of the head in ONE LDS buffer; thread = (row tid / 8, slice tid % 8).
template <int D>
__global__ __launch_bounds__(256) void sdpa_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ a, float* __restrict__ out, float* __restrict__ Pm,
                                                       int N, int heads, float scale) {
    extern __shared__ float sm[];
    float* sKV = sm;                         // [N][D+1]: K, later V
    float* sQ = sKV + (long)N * (D + 1);     // [32][D+1]
    float* sP = sQ + 32 * (D + 1);           // [32][N+1]
    const int C = heads * D, b = blockIdx.z, h = blockIdx.y, r0 = blockIdx.x * 32, tid = threadIdx.x;
    const float* base = qkv + (long)b * N * 3 * C + h * D;
    for (int i = tid; i < N * D; i += 256) { const int n = i / D, d = i % D; sKV[n * (D + 1) + d] = base[(long)n * 3 * C + C + d]; }
    for (int i = tid; i < 32 * D; i += 256) { const int r = i / D, d = i % D; sQ[r * (D + 1) + d] = (r0 + r < N) ? base[(long)(r0 + r) * 3 * C + d] : 0.f; }
    __syncthreads();
    const int r = tid >> 3, j = tid & 7;
    float mx = -INFINITY;
    for (int n = j; n < N; n += 8) {
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) s = fmaf(sQ[r * (D + 1) + d], sKV[n * (D + 1) + d], s);
        s *= scale;
        sP[r * (N + 1) + n] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int n = j; n < N; n += 8) { const float e = __expf(sP[r * (N + 1) + n] - mx); sP[r * (N + 1) + n] = e; sum += e; }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.f / sum;
    for (int n = j; n < N; n += 8) {
        const float pv = sP[r * (N + 1) + n] * inv;
        sP[r * (N + 1) + n] = pv;
        if (r0 + r < N) Pm[(((long)b * heads + h) * N + r0 + r) * N + n] = pv;
    }
    __syncthreads();                          // everybody is done with K
    for (int i = tid; i < N * D; i += 256) { const int n = i / D, d = i % D; sKV[n * (D + 1) + d] = base[(long)n * 3 * C + 2 * C + d]; }
    __syncthreads();
    if (r0 + r < N)
        for (int d = j; d < D; d += 8) {
            float o = 0.f;
            for (int n = 0; n < N; ++n) o = fmaf(sP[r * (N + 1) + n], sKV[n * (D + 1) + d], o);
            const float av = a ? a[(long)b * C + h * D + d] : 1.f;
            out[((long)b * N + r0 + r) * C + h * D + d] = av * o;
        }
}
// backward, rows: (b, head, 32 query rows):  dO = a g;  dP = dO V^T;  dS = scale P (dP - rowsum(dP P));  dQ = dS K;  dS -> global
template <int D>
__global__ __launch_bounds__(256) void sdpa_bwd_rows_kernel(const float* __restrict__ g, const float* __restrict__ qkv, const float* __restrict__ Pm,
                                                            const float* __restrict__ a, float* __restrict__ dqkv, float* __restrict__ dSm,
                                                            int N, int heads, float scale) {
    extern __shared__ float sm[];
    float* sKV = sm;                         // [N][D+1]: V, later K
    float* sdO = sKV + (long)N * (D + 1);    // [32][D+1]
    float* sP = sdO + 32 * (D + 1);          // [32][N+1]: P, then dS
    const int C = heads * D, b = blockIdx.z, h = blockIdx.y, r0 = blockIdx.x * 32, tid = threadIdx.x;
    const float* base = qkv + (long)b * N * 3 * C + h * D;
    for (int i = tid; i < N * D; i += 256) { const int n = i / D, d = i % D; sKV[n * (D + 1) + d] = base[(long)n * 3 * C + 2 * C + d]; }
    for (int i = tid; i < 32 * D; i += 256) {
        const int rr = i / D, d = i % D;
        sdO[rr * (D + 1) + d] = (r0 + rr < N) ? g[((long)b * N + r0 + rr) * C + h * D + d] * (a ? a[(long)b * C + h * D + d] : 1.f) : 0.f;
    }
    for (int i = tid; i < 32 * N; i += 256) { const int rr = i / N, n = i % N; sP[rr * (N + 1) + n] = (r0 + rr < N) ? Pm[(((long)b * heads + h) * N + r0 + rr) * N + n] : 0.f; }
    __syncthreads();
    const int r = tid >> 3, j = tid & 7;
    float rd = 0.f;
    for (int n = j; n < N; n += 8) {
        float s = 0.f;
#pragma unroll 8
        for (int d = 0; d < D; ++d) s = fmaf(sdO[r * (D + 1) + d], sKV[n * (D + 1) + d], s);
        rd = fmaf(s, sP[r * (N + 1) + n], rd);
        // park dP in the global dS buffer (read back below by the same thread)
        if (r0 + r < N) dSm[(((long)b * heads + h) * N + r0 + r) * N + n] = s;
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) rd += __shfl_xor(rd, o, 64);
    for (int n = j; n < N; n += 8) {
        const long o = (((long)b * heads + h) * N + r0 + r) * N + n;
        const float ds = (r0 + r < N) ? sP[r * (N + 1) + n] * (dSm[o] - rd) * scale : 0.f;
        sP[r * (N + 1) + n] = ds;
        if (r0 + r < N) dSm[o] = ds;
    }
    __syncthreads();                          // dS complete; everybody is done with V
    for (int i = tid; i < N * D; i += 256) { const int n = i / D, d = i % D; sKV[n * (D + 1) + d] = base[(long)n * 3 * C + C + d]; }
    __syncthreads();
    if (r0 + r < N)
        for (int d = j; d < D; d += 8) {
            float acc = 0.f;
            for (int n = 0; n < N; ++n) acc = fmaf(sP[r * (N + 1) + n], sKV[n * (D + 1) + d], acc);
            dqkv[((long)b * N + r0 + r) * 3 * C + h * D + d] = acc;
        }
}
// backward, keys: (b, head, 32 keys):  dV[n] = sum_r P[r][n] dO[r];  dK[n] = sum_r dS[r][n] Q[r];   e[b, c] = sum_n g out  (key block 0)
template <int D>
__global__ __launch_bounds__(256) void sdpa_bwd_keys_kernel(const float* __restrict__ g, const float* __restrict__ qkv, const float* __restrict__ Pm,
                                                            const float* __restrict__ dSm, const float* __restrict__ outp, const float* __restrict__ a,
                                                            float* __restrict__ dqkv, float* __restrict__ e_out, int N, int heads) {
    extern __shared__ float sm[];
    float* sX = sm;                          // [N][D+1]: dO, later Q
    float* sT = sX + (long)N * (D + 1);      // [N rows][33]: P[:, n0..n0+31], later dS
    const int C = heads * D, b = blockIdx.z, h = blockIdx.y, n0 = blockIdx.x * 32, tid = threadIdx.x;
    const float* base = qkv + (long)b * N * 3 * C + h * D;
    for (int i = tid; i < N * D; i += 256) {
        const int rr = i / D, d = i % D;
        sX[rr * (D + 1) + d] = g[((long)b * N + rr) * C + h * D + d] * (a ? a[(long)b * C + h * D + d] : 1.f);
    }
    for (int i = tid; i < N * 32; i += 256) { const int rr = i / 32, c = i % 32; sT[rr * 33 + c] = (n0 + c < N) ? Pm[(((long)b * heads + h) * N + rr) * N + n0 + c] : 0.f; }
    __syncthreads();
    const int nn = tid >> 3, j = tid & 7;     // key n0 + nn, channels j, j+8, ...
    if (n0 + nn < N)
        for (int d = j; d < D; d += 8) {
            float acc = 0.f;
            for (int rr = 0; rr < N; ++rr) acc = fmaf(sT[rr * 33 + nn], sX[rr * (D + 1) + d], acc);
            dqkv[((long)b * N + n0 + nn) * 3 * C + 2 * C + h * D + d] = acc;
        }
    if (e_out && blockIdx.x == 0 && tid < D) {
        float acc = 0.f;
        for (int rr = 0; rr < N; ++rr) { const long o = ((long)b * N + rr) * C + h * D + tid; acc = fmaf(g[o], outp[o], acc); }
        e_out[(long)b * C + h * D + tid] = acc;
    }
    __syncthreads();
    for (int i = tid; i < N * D; i += 256) { const int rr = i / D, d = i % D; sX[rr * (D + 1) + d] = base[(long)rr * 3 * C + d]; }
    for (int i = tid; i < N * 32; i += 256) { const int rr = i / 32, c = i % 32; sT[rr * 33 + c] = (n0 + c < N) ? dSm[(((long)b * heads + h) * N + rr) * N + n0 + c] : 0.f; }
    __syncthreads();
    if (n0 + nn < N)
        for (int d = j; d < D; d += 8) {
            float acc = 0.f;
            for (int rr = 0; rr < N; ++rr) acc = fmaf(sT[rr * 33 + nn], sX[rr * (D + 1) + d], acc);
            dqkv[((long)b * N + n0 + nn) * 3 * C + C + h * D + d] = acc;
        }
}

// ---- structure_loss (multi_train_TransFuse.py:29-38) ----------------------------------------------------------------------------
// weit = 1 + 5 |avg_pool2d(mask, 31, stride 1, pad 15) - mask|   (count_include_pad: always / 961).  Separable running sums would be
// faster; the mask is 256x256 and this runs once per domain per step.
__global__ __launch_bounds__(256) void box31_rows_kernel(const float* __restrict__ m, float* __restrict__ tmp, int B, int H, int W) {
    const long total = (long)B * H * W;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int w = (int)(e % W); const long row = e / W;
        float s = 0.f;
        for (int k = -15; k <= 15; ++k) { const int ww = w + k; if (ww >= 0 && ww < W) s += m[row * W + ww]; }
        tmp[e] = s;
    }
}
__global__ __launch_bounds__(256) void box31_cols_kernel(const float* __restrict__ tmp, const float* __restrict__ m, float* __restrict__ weit, int B, int H, int W) {
    const long total = (long)B * H * W;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int w = (int)(e % W), h = (int)((e / W) % H); const long b = e / ((long)W * H);
        float s = 0.f;
        for (int k = -15; k <= 15; ++k) { const int hh = h + k; if (hh >= 0 && hh < H) s += tmp[(b * H + hh) * W + w]; }
        weit[e] = 1.f + 5.f * fabsf(s * (1.f / 961.f) - m[e]);
    }
}
// per sample: S0 = sum weit*bce, S1 = sum weit, S2 = sum p*m*weit, S3 = sum (p+m)*weit   (double).  gridDim.y workgroups share a sample
// (a domain batch has 8 samples: one workgroup per sample left 248 CUs idle for 80 us) and add their partial sums with double atomics
// into the zeroed sums (the addition order moves the result by ~1e-16 relative, far below the float it is rounded to)
__global__ __launch_bounds__(256) void sl_sums_kernel(const float* __restrict__ pred, const float* __restrict__ mask, const float* __restrict__ weit,
                                                      double* __restrict__ sums, long HW) {
    __shared__ double red[4][256];
    const long b = blockIdx.x;
    const long chunk = (HW + gridDim.y - 1) / gridDim.y, e0 = blockIdx.y * chunk, e1 = min(HW, e0 + chunk);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const float x = pred[b * HW + e], m = mask[b * HW + e], w = weit[b * HW + e];
        // binary_cross_entropy_with_logits: max(x,0) - x*m + log(1 + exp(-|x|))
        const float bce = fmaxf(x, 0.f) - x * m + log1pf(__expf(-fabsf(x)));
        const float p = sigmoidf_(x);
        s0 += (double)(w * bce); s1 += (double)w; s2 += (double)(p * m * w); s3 += (double)((p + m) * w);
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2; red[3][threadIdx.x] = s3;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x < 4) atomicAdd(&sums[b * 4 + threadIdx.x], red[threadIdx.x][0]);
}
// loss = mean_b [ S0/S1 + 1 - (S2 + 1) / (S3 - S2 + 1) ]
__global__ void sl_final_kernel(const double* __restrict__ sums, float* __restrict__ loss, int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int b = 0; b < B; ++b) { const double* s = sums + 4 * b; t += s[0] / s[1] + 1.0 - (s[2] + 1.0) / (s[3] - s[2] + 1.0); }
        loss[0] = (float)(t / (double)B);
    }
}
// d loss / d x = gscale/B * [ w (p - m) / S1  -  ( dinter (U + 1) - (I + 1) (dunion - dinter) ) / (U - I + 1)^2 ],  dinter = w m p(1-p), dunion = w p(1-p)
__global__ __launch_bounds__(256) void sl_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ mask, const float* __restrict__ weit,
                                                     const double* __restrict__ sums, const float* __restrict__ gscale, float* __restrict__ dpred, long HW, int B) {
    const long total = (long)B * HW;
    const float gs = gscale[0] / (float)B;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = e / HW;
        const double* s = sums + 4 * b;
        const float S1 = (float)s[1], I = (float)s[2], U = (float)s[3];
        const float x = pred[e], m = mask[e], w = weit[e];
        const float p = sigmoidf_(x), dp = p * (1.f - p);
        const float den = U - I + 1.f;
        const float dI = w * m * dp, dU = w * dp;
        const float diou = -(dI * den - (I + 1.f) * (dU - dI)) / (den * den);
        dpred[e] = gs * (w * (p - m) / S1 + diou);
    }
}

}  // namespace

#define TF_LAUNCH(kernel, grid, block, smem, s, ...) do { hipLaunchKernelGGL(kernel, dim3 grid, dim3(block), smem, s, __VA_ARGS__); MDVIT_LAUNCH_CHECK(); } while (0)

extern "C" int mdvit_imgconv_fwd(const float* img, const float* w, float* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, void* stream) {
    MDVIT_CHECK_ARG(img && w && y && Cin == 3 && ksize == 7, MDVIT_E_SHAPE, "imgconv_fwd: built for in_chans == 3, kernel 7 (got %d, %d)", Cin, ksize);
    MDVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 4 == 0 && Cout <= 64, MDVIT_E_SHAPE, "imgconv_fwd: bad shape");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    TF_LAUNCH((imgconv_fwd_kernel<3, 7>), (tf_grid((long)B * Ho * Wo * Cout / 4)), 256, sizeof(float) * 147 * Cout, (hipStream_t)stream, img, w, y, B, H, W, Cout);
    return MDVIT_OK;
}
extern "C" int mdvit_imgconv_wgrad(const float* img, const float* dy, float* dw, void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                   int32_t Cout, int32_t ksize, int32_t accumulate, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(img && dy && dw && Cin == 3 && ksize == 7, MDVIT_E_SHAPE, "imgconv_wgrad: built for in_chans == 3, kernel 7");
    MDVIT_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout <= 64, MDVIT_E_SHAPE, "imgconv_wgrad: bad shape");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long npix = (long)B * Ho * Wo;
    const int ppb = (int)max(64L, (npix + 1023) / 1024);
    const int nblk = cdiv(npix, ppb);
    MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nblk, 147 * Cout, "imgconv_wgrad");
    TF_LAUNCH((imgconv_wgrad_kernel<3, 7>), (nblk), 256, 0, s, img, dy, (float*)ws, B, H, W, Cout, ppb);
    return mdvit_reduce_partials((const float*)ws, nblk, 147L * Cout, 147 * Cout, dw, 0, nullptr, accumulate, s);
}
extern "C" int mdvit_imgconv_im2col(const float* img, float* col, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t ksize, int32_t ldc, void* stream) {
    MDVIT_CHECK_ARG(img && col && Cin == 3 && ksize == 7 && ldc >= 147 && ldc % 4 == 0, MDVIT_E_SHAPE, "imgconv_im2col: built for in_chans == 3, kernel 7, ldc >= 147 and %% 4 == 0");
    MDVIT_CHECK_ARG(B > 0 && H > 0 && W > 0, MDVIT_E_SHAPE, "imgconv_im2col: bad shape");
    const long n = (long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * ldc;
    TF_LAUNCH((imgconv_im2col_kernel<3, 7>), (tf_grid(n)), 256, 0, (hipStream_t)stream, img, col, B, H, W, ldc);
    return MDVIT_OK;
}
extern "C" int mdvit_maxpool3x3s2_fwd(const float* x, float* y, void* idx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    MDVIT_CHECK_ARG(x && y && idx && B > 0 && H > 0 && W > 0 && C > 0, MDVIT_E_SHAPE, "maxpool_fwd: bad arguments");
    TF_LAUNCH(maxpool_fwd_kernel, (tf_grid((long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * C)), 256, 0, (hipStream_t)stream, x, y, (uint8_t*)idx, B, H, W, C);
    return MDVIT_OK;
}
extern "C" int mdvit_maxpool3x3s2_bwd(const float* dy, const void* idx, float* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    MDVIT_CHECK_ARG(dy && dx && idx && B > 0 && H > 0 && W > 0 && C > 0, MDVIT_E_SHAPE, "maxpool_bwd: bad arguments");
    TF_LAUNCH(maxpool_bwd_kernel, (tf_grid((long)B * H * W * C)), 256, 0, (hipStream_t)stream, dy, (const uint8_t*)idx, dx, B, H, W, C);
    return MDVIT_OK;
}
extern "C" int mdvit_resize_ac_fwd(const float* x, float* y, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C, void* stream) {
    MDVIT_CHECK_ARG(x && y && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, MDVIT_E_SHAPE, "resize_ac_fwd: bad arguments");
    TF_LAUNCH(resize_ac_fwd_kernel, (tf_grid((long)B * Ho * Wo * C)), 256, 0, (hipStream_t)stream, x, y, B, Hi, Wi, Ho, Wo, C);
    return MDVIT_OK;
}
extern "C" int mdvit_resize_ac_bwd(const float* dy, float* dx, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, MDVIT_E_SHAPE, "resize_ac_bwd: bad arguments");
    TF_LAUNCH(resize_ac_bwd_kernel, (tf_grid((long)B * Hi * Wi * C)), 256, 0, s, dy, dx, B, Hi, Wi, Ho, Wo, C);
    return MDVIT_OK;
}
extern "C" int mdvit_ew(const float* a, const float* b, float* y, int64_t n, int32_t mode, void* stream) {
    MDVIT_CHECK_ARG(a && y && n > 0 && mode >= 0 && mode <= 3 && (b || mode == 0), MDVIT_E_SHAPE, "ew: bad arguments");
    TF_LAUNCH(ew_kernel, (tf_grid(n)), 256, 0, (hipStream_t)stream, a, b, y, (long)n, mode);
    return MDVIT_OK;
}
extern "C" int mdvit_add3(const float* a, const float* b, const float* c, float* y, int64_t n, void* stream) {
    MDVIT_CHECK_ARG(a && b && c && y && n > 0, MDVIT_E_SHAPE, "add3: bad arguments");
    const bool vec = aligned16(a) && aligned16(b) && aligned16(c) && aligned16(y);
    const long n4 = vec ? (long)n / 4 : 0;
    TF_LAUNCH(add3_kernel, (tf_grid(vec ? (long)n / 4 + 3 : (long)n)), 256, 0, (hipStream_t)stream, a, b, c, y, n4, (long)n);
    return MDVIT_OK;
}
extern "C" int mdvit_add_parts(const float* a, const float* b, const void* const* parts, int32_t G, int64_t part_elems, float* y, void* stream) {
    MDVIT_CHECK_ARG(a && parts && y && G >= 1 && G <= 8 && part_elems > 0 && part_elems % 4 == 0, MDVIT_E_SHAPE, "add_parts: 1 <= G <= 8 parts of a multiple of 4 elements");
    MDVIT_CHECK_ARG(aligned16(a) && (!b || aligned16(b)) && aligned16(y), MDVIT_E_ALIGN, "add_parts: operands must be 16-byte aligned");
    AddPartsArgs pa; memset(&pa, 0, sizeof(pa));
    for (int g = 0; g < G; ++g) { MDVIT_CHECK_ARG(parts[g] && aligned16(parts[g]), MDVIT_E_ALIGN, "add_parts: part %d null or unaligned", g); pa.p[g] = (const float*)parts[g]; }
    const long part4 = (long)part_elems / 4, n4 = part4 * G;
    TF_LAUNCH(add_parts_kernel, (tf_grid(n4)), 256, 0, (hipStream_t)stream, a, b, pa, y, part4, n4);
    return MDVIT_OK;
}
extern "C" int mdvit_add_bcast(const float* x, const float* pe, float* y, int32_t B, int64_t R, void* stream) {
    MDVIT_CHECK_ARG(x && pe && y && B > 0 && R > 0, MDVIT_E_SHAPE, "add_bcast: bad arguments");
    TF_LAUNCH(add_bcast_kernel, (tf_grid((long)B * R)), 256, 0, (hipStream_t)stream, x, pe, y, (long)R, (long)B * R);
    return MDVIT_OK;
}
extern "C" int mdvit_sum_batch(const float* g, float* out, int32_t B, int64_t R, void* stream) {
    MDVIT_CHECK_ARG(g && out && B > 0 && R > 0, MDVIT_E_SHAPE, "sum_batch: bad arguments");
    TF_LAUNCH(sum_batch_kernel, (tf_grid((long)R)), 256, 0, (hipStream_t)stream, g, out, B, (long)R);
    return MDVIT_OK;
}
extern "C" int mdvit_gate_fwd(const float* x, const float* s, float* y, int32_t B, int64_t P, int32_t C, int32_t mode, void* stream) {
    MDVIT_CHECK_ARG(x && s && y && B > 0 && P > 0 && C > 0 && (mode == 0 || mode == 1), MDVIT_E_SHAPE, "gate_fwd: bad arguments");
    TF_LAUNCH(gate_fwd_kernel, (tf_grid((long)B * P * C)), 256, 0, (hipStream_t)stream, x, s, y, (long)P, C, (long)B * P * C, mode);
    return MDVIT_OK;
}
extern "C" size_t mdvit_gate_bwd_ws_bytes(int32_t B, int64_t P, int32_t C, int32_t mode) {
    return mode == 1 ? sizeof(float) * (size_t)B * (size_t)cdiv(P, 64) * C : 0;
}
extern "C" int mdvit_gate_bwd(const float* g, const float* x, const float* s, float* dx, float* ds, void* ws, size_t ws_bytes, int32_t B, int64_t P, int32_t C,
                              int32_t mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    MDVIT_CHECK_ARG(g && x && s && dx && ds && B > 0 && P > 0 && C > 0 && (mode == 0 || mode == 1), MDVIT_E_SHAPE, "gate_bwd: bad arguments");
    if (mode == 0) { TF_LAUNCH(gate_bwd_spatial_kernel, (tf_grid((long)B * P, 4)), 256, 0, st, g, x, s, dx, ds, (long)B * P, C); return MDVIT_OK; }
    const int nchunk = cdiv(P, 64);
    MDVIT_CHECK_ARG(ws && ws_bytes >= mdvit_gate_bwd_ws_bytes(B, P, C, 1), MDVIT_E_WORKSPACE, "gate_bwd: workspace too small (mdvit_gate_bwd_ws_bytes)");
    TF_LAUNCH(gate_bwd_channel_kernel, (cdiv(C, 64), nchunk, B), 256, 0, st, g, x, s, dx, (float*)ws, (long)P, C, 64);
    TF_LAUNCH(gate_bwd_channel_final_kernel, (cdiv(B * C, 256)), 256, 0, st, (const float*)ws, s, ds, B, C, nchunk);
    return MDVIT_OK;
}
extern "C" int mdvit_chanpool_fwd(const float* x, float* y, int32_t* idx, int64_t M, int32_t C, void* stream) {
    MDVIT_CHECK_ARG(x && y && idx && M > 0 && C > 0, MDVIT_E_SHAPE, "chanpool_fwd: bad arguments");
    TF_LAUNCH(chanpool_fwd_kernel, (tf_grid((long)M, 4)), 256, 0, (hipStream_t)stream, x, y, idx, (long)M, C);
    return MDVIT_OK;
}
extern "C" int mdvit_chanpool_bwd(const float* dy, const int32_t* idx, float* dx, int64_t M, int32_t C, void* stream) {
    MDVIT_CHECK_ARG(dy && dx && idx && M > 0 && C > 0, MDVIT_E_SHAPE, "chanpool_bwd: bad arguments");
    TF_LAUNCH(chanpool_bwd_kernel, (tf_grid((long)M * C)), 256, 0, (hipStream_t)stream, dy, idx, dx, (long)M, C);
    return MDVIT_OK;
}
extern "C" int mdvit_conv7x7_2to1_fwd(const float* x, const float* w, float* y, int32_t B, int32_t H, int32_t W, void* stream) {
    MDVIT_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0, MDVIT_E_SHAPE, "conv7x7_2to1_fwd: bad arguments");
    TF_LAUNCH(conv7_fwd_kernel, (tf_grid((long)B * H * W)), 256, 0, (hipStream_t)stream, x, w, y, B, H, W);
    return MDVIT_OK;
}
extern "C" int mdvit_conv7x7_2to1_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, int32_t B, int32_t H, int32_t W, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(dy && x && w && B > 0 && H > 0 && W > 0, MDVIT_E_SHAPE, "conv7x7_2to1_bwd: bad arguments");
    if (dx) TF_LAUNCH(conv7_dgrad_kernel, (tf_grid((long)B * H * W)), 256, 0, s, dy, w, dx, B, H, W);
    if (dw) TF_LAUNCH(conv7_wgrad_kernel, (98), 256, 0, s, dy, x, dw, B, H, W);
    return MDVIT_OK;
}
extern "C" int mdvit_bn1_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* num_batches_tracked,
                             float* y, float* stat, int64_t M, int32_t groups, int32_t training, float eps, float momentum, void* stream) {
    MDVIT_CHECK_ARG(x && gamma && beta && running_mean && running_var && y && stat && M > 0 && groups > 0 && M % groups == 0, MDVIT_E_SHAPE,
                    "bn1_fwd: bad arguments (M=%ld groups=%d)", (long)M, groups);
    TF_LAUNCH(bn1_fwd_kernel, (1), 1024, 0, (hipStream_t)stream, x, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, y, stat, (long)M, groups,
              training, eps, momentum);
    return MDVIT_OK;
}
extern "C" int mdvit_bn1_bwd(const float* g, const float* x, const float* gamma, const float* stat, float* dx, float* dgamma_dbeta, int64_t M, int32_t groups,
                             int32_t training, void* stream) {
    MDVIT_CHECK_ARG(g && x && gamma && stat && dx && dgamma_dbeta && M > 0 && groups > 0 && M % groups == 0, MDVIT_E_SHAPE, "bn1_bwd: bad arguments");
    TF_LAUNCH(bn1_bwd_kernel, (1), 1024, 0, (hipStream_t)stream, g, x, gamma, stat, dx, dgamma_dbeta, (long)M, groups, training);
    return MDVIT_OK;
}
extern "C" int mdvit_subsample2(const float* src, float* dst, int32_t B, int32_t H, int32_t W, int32_t C, int32_t backward, void* stream) {
    MDVIT_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && C > 0, MDVIT_E_SHAPE, "subsample2: bad arguments");
    const long n = backward ? (long)B * H * W * C : (long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * C;
    TF_LAUNCH(subsample2_kernel, (tf_grid(n)), 256, 0, (hipStream_t)stream, src, dst, B, H, W, C, backward);
    return MDVIT_OK;
}
extern "C" int mdvit_patchify(const float* img, float* out, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t patch, void* stream) {
    MDVIT_CHECK_ARG(img && out && B > 0 && Cin > 0 && patch > 0 && H % patch == 0 && W % patch == 0, MDVIT_E_SHAPE, "patchify: H, W must be multiples of the patch size");
    TF_LAUNCH(patchify_kernel, (tf_grid((long)B * Cin * H * W)), 256, 0, (hipStream_t)stream, img, out, B, Cin, H, W, patch);
    return MDVIT_OK;
}
extern "C" int mdvit_dropout2d(const float* x, float* y, int32_t B, int64_t P, int32_t C, float p, uint32_t key0, uint32_t key1, const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(x && y && B > 0 && P > 0 && C > 0 && p >= 0.f && p < 1.f, MDVIT_E_SHAPE, "dropout2d: bad arguments");
    TF_LAUNCH(dropout2d_kernel, (tf_grid((long)B * P * C)), 256, 0, (hipStream_t)stream, x, y, (long)P, C, (long)B * P * C, key0, key1,
              mdvit_drop_thresh(p), 1.f / (1.f - p), drop_seed);
    return MDVIT_OK;
}
static int tf_set_lds(const void* k, int& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 31) dev = 0;
    if (mask & (1 << dev)) return MDVIT_OK;
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "sdpa: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    mask |= 1 << dev;
    return MDVIT_OK;
}
extern "C" int mdvit_sdpa_fwd(const float* qkv, const float* a, float* out, float* P, int32_t B, int32_t N, int32_t C, int32_t heads, void* stream) {
    MDVIT_CHECK_ARG(qkv && out && P && B > 0 && N > 0 && N <= 256 && heads > 0 && C == heads * 64, MDVIT_E_SHAPE,
                    "sdpa_fwd: built for head dimension 64 and N <= 256 (N=%d C=%d heads=%d)", N, C, heads);
    const size_t smem = sizeof(float) * ((size_t)N * 65 + 32 * 65 + 32 * (size_t)(N + 1));
    static int mask = 0;
    const int rc = tf_set_lds(reinterpret_cast<const void*>(&sdpa_fwd_kernel<64>), mask);
    if (rc != MDVIT_OK) return rc;
    TF_LAUNCH((sdpa_fwd_kernel<64>), (cdiv(N, 32), heads, B), 256, smem, (hipStream_t)stream, qkv, a, out, P, N, heads, 0.125f);
    return MDVIT_OK;
}
/* dS: scratch [B, heads, N, N] floats */
extern "C" int mdvit_sdpa_bwd(const float* g, const float* qkv, const float* P, const float* out, const float* a, float* dqkv, float* e, float* dS, int32_t B,
                              int32_t N, int32_t C, int32_t heads, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(g && qkv && P && out && dqkv && dS && B > 0 && N > 0 && N <= 256 && heads > 0 && C == heads * 64, MDVIT_E_SHAPE,
                    "sdpa_bwd: built for head dimension 64 and N <= 256 (N=%d C=%d heads=%d)", N, C, heads);
    MDVIT_CHECK_ARG((a == nullptr) == (e == nullptr), MDVIT_E_SHAPE, "sdpa_bwd: the adapter scale a and its gradient carrier e go together");
    static int m1 = 0, m2 = 0;
    int rc = tf_set_lds(reinterpret_cast<const void*>(&sdpa_bwd_rows_kernel<64>), m1);
    if (rc == MDVIT_OK) rc = tf_set_lds(reinterpret_cast<const void*>(&sdpa_bwd_keys_kernel<64>), m2);
    if (rc != MDVIT_OK) return rc;
    const size_t smem1 = sizeof(float) * ((size_t)N * 65 + 32 * 65 + 32 * (size_t)(N + 1));
    const size_t smem2 = sizeof(float) * ((size_t)N * 65 + (size_t)N * 33);
    TF_LAUNCH((sdpa_bwd_rows_kernel<64>), (cdiv(N, 32), heads, B), 256, smem1, s, g, qkv, P, a, dqkv, dS, N, heads, 0.125f);
    TF_LAUNCH((sdpa_bwd_keys_kernel<64>), (cdiv(N, 32), heads, B), 256, smem2, s, g, qkv, P, dS, out, a, dqkv, e, N, heads);
    return MDVIT_OK;
}
extern "C" int mdvit_structure_weight(const float* mask, float* tmp, float* weit, int32_t B, int32_t H, int32_t W, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(mask && tmp && weit && B > 0 && H > 0 && W > 0, MDVIT_E_SHAPE, "structure_weight: bad arguments");
    TF_LAUNCH(box31_rows_kernel, (tf_grid((long)B * H * W)), 256, 0, s, mask, tmp, B, H, W);
    TF_LAUNCH(box31_cols_kernel, (tf_grid((long)B * H * W)), 256, 0, s, tmp, mask, weit, B, H, W);
    return MDVIT_OK;
}
extern "C" int mdvit_structure_loss_fwd(const float* pred, const float* mask, const float* weit, double* sums, float* loss, int32_t B, int64_t HW, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(pred && mask && weit && sums && loss && B > 0 && HW > 0, MDVIT_E_SHAPE, "structure_loss_fwd: bad arguments");
    MDVIT_ZERO(sums, sizeof(double) * 4 * (size_t)B, s);
    TF_LAUNCH(sl_sums_kernel, (B, (int)min((HW + 4095) / 4096, 64L)), 256, 0, s, pred, mask, weit, sums, (long)HW);
    TF_LAUNCH(sl_final_kernel, (1), 64, 0, s, sums, loss, B);
    return MDVIT_OK;
}
extern "C" int mdvit_structure_loss_bwd(const float* pred, const float* mask, const float* weit, const double* sums, const float* gscale, float* dpred, int32_t B,
                                        int64_t HW, void* stream) {
    MDVIT_CHECK_ARG(pred && mask && weit && sums && gscale && dpred && B > 0 && HW > 0, MDVIT_E_SHAPE, "structure_loss_bwd: bad arguments");
    TF_LAUNCH(sl_bwd_kernel, (tf_grid((long)B * HW)), 256, 0, (hipStream_t)stream, pred, mask, weit, sums, gscale, dpred, (long)HW, B);
    return MDVIT_OK;
}
