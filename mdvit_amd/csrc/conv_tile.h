// LDS-tiled depthwise window convolutions (3x3 / 5x5 / 7x7, stride 1, zero padding) on NHWC token images, shared by the
// attention's ConvRelPosEnc (attn.hip) and the stride-1 depthwise 3x3 convolutions (ConvPosEnc, DWCPatchEmbed: conv.hip).
// Every translation unit that includes this header gets its own copy in its anonymous namespace.
#pragma once
#include "common.h"

namespace {

struct CtGeom { int B, H, W; };

// ---- LDS-tiled depthwise window convolution over the token image -----------------------------------
// y[b,n,cy] = bias[ci] + sum_{i,j} w[ci][i][j] * x[b, n + (i-R, j-R), cx]      (FLIP: w[ci][WIN-1-i][WIN-1-j], no bias)
// for the ncls channels of ONE window class (ci = class-local index).  Block = 8 x 16 token tile x 32 channels:
// the tile plus halo sits in LDS ([row][col][channel], channel fastest -> conflict-free), thread (channel, row)
// produces 16 outputs along w from WIN input rows held in registers (16*WIN FMAs per TW+WIN-1 LDS reads).
constexpr int CT_TH = 8, CT_TW = 16, CT_CL = 32;

// Stage an LHxLW token window x 32 channels into LDS ([position][channel]); zero outside the image / past `nch` channels.
// 8 lanes x float4 cover one position, 32 positions per pass; ALL global loads of the window are issued before the
// first LDS store (a load -> store loop would pay the HBM latency once per iteration).  x already points at the first
// channel of the block; nch % 4 == 0.
template <int LH, int LW>
struct CtWindow {
    static constexpr int NP = LH * LW, PASSES = (NP + 31) / 32;
    float4 v[PASSES];
    // request the whole window (global -> registers); nothing waits here
    __device__ __forceinline__ void load(const float* __restrict__ x, long ldx, long img, int h0, int w0, int H, int W, int nch) {
        const int q4 = (threadIdx.x & 7) * 4, pl = threadIdx.x >> 3;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int p = pl + 32 * i;
            const int hh = h0 + p / LW, ww = w0 + p % LW;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < NP && hh >= 0 && hh < H && ww >= 0 && ww < W && q4 < nch)
                v[i] = *reinterpret_cast<const float4*>(x + (img + (long)hh * W + ww) * ldx + q4);
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ sx) const {
        const int q4 = (threadIdx.x & 7) * 4, pl = threadIdx.x >> 3;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int p = pl + 32 * i;
            if (p < NP) *reinterpret_cast<float4*>(sx + p * CT_CL + q4) = v[i];
        }
    }
};

template <int LH, int LW>
__device__ __forceinline__ void ct_stage_window(float* __restrict__ sx, const float* __restrict__ x, long ldx, long img,
                                                int h0, int w0, int H, int W, int nch) {
    CtWindow<LH, LW> win;
    win.load(x, ldx, img, h0, w0, H, W, nch);
    win.store(sx);
}

// body of one (tile, 32-channel block, image) workgroup; sx / sw: LDS of at least (8+2R)(16+2R)*32 and 32*WIN*WIN floats
// GD = 2: output channel c reads INPUT channel c / 2 (the data gradient of the grouped decoder convolution: one dy channel feeds the two concat channels of its group);
// PAIR: the outputs of channels 2 g, 2 g + 1 are added and stored as channel g (its forward: a depthwise pass over the concat channels + the group sum).  Both need
// ncls % 32 == 0.
template <int WIN, bool FLIP, int GD = 1, bool PAIR = false>
__device__ __forceinline__ void conv_tile_body(float* __restrict__ sx, float* __restrict__ sw, int cblock,
                                               const float* __restrict__ x, long ldx, int xoff,
                                               const float* __restrict__ w, const float* __restrict__ bias,
                                               float* __restrict__ y, long ldy, int yoff,
                                               int H, int W, int ncls, int tiles_w, int add_center) {
    constexpr int R = WIN / 2, LH = CT_TH + 2 * R, LW = CT_TW + 2 * R;
    const int b = blockIdx.z, c0 = cblock * CT_CL;
    const int th0 = (blockIdx.x / tiles_w) * CT_TH, tw0 = (blockIdx.x % tiles_w) * CT_TW;
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const long img = (long)b * H * W;
    // the window AND the 32 x WIN^2 weights are requested before anything is stored to LDS: one exposed load latency per workgroup
    // (a load -> LDS-store loop over the weights paid it up to seven more times)
    CtWindow<LH, LW> win;
    win.load(x + xoff + c0 / GD, ldx, img, th0 - R, tw0 - R, H, W, (ncls - c0) / GD);
    constexpr int NWQ = (CT_CL * WIN * WIN + 255) / 256;
    float wq[NWQ];
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
        const int i = threadIdx.x + 256 * j, c = i / (WIN * WIN), t = i % (WIN * WIN);
        const bool ok = i < CT_CL * WIN * WIN && c0 + c < ncls;
        const float wv = w[ok ? (long)(c0 + c) * WIN * WIN + (FLIP ? WIN * WIN - 1 - t : t) : 0];
        wq[j] = ok ? wv : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < CT_CL * WIN * WIN) sw[i] = wq[j];
    }
    win.store(sx);
    __syncthreads();
    const int h = th0 + rl;
    if (c0 + cl >= ncls || h >= H) return;
    float acc[CT_TW];
    const float b0 = (!FLIP && bias) ? bias[c0 + cl] : 0.f;
#pragma unroll
    for (int t = 0; t < CT_TW; ++t) acc[t] = b0;
#pragma unroll
    for (int i = 0; i < WIN; ++i) {
        float row[LW], wr[WIN];
#pragma unroll
        for (int t = 0; t < LW; ++t) row[t] = sx[((rl + i) * LW + t) * CT_CL + cl / GD];
        if (i == R && add_center) {          // y = x + conv(x)  (ConvPosEnc) / dx = g + conv^T(g)
#pragma unroll
            for (int t = 0; t < CT_TW; ++t) acc[t] += row[t + R];
        }
#pragma unroll
        for (int j = 0; j < WIN; ++j) wr[j] = sw[cl * WIN * WIN + i * WIN + j];
#pragma unroll
        for (int t = 0; t < CT_TW; ++t)
#pragma unroll
            for (int j = 0; j < WIN; ++j) acc[t] = fmaf(wr[j], row[t + j], acc[t]);
    }
    if constexpr (PAIR) {
#pragma unroll
        for (int t = 0; t < CT_TW; ++t) {
            const float s2 = acc[t] + __shfl_xor(acc[t], 1);
            if (!(cl & 1) && tw0 + t < W) y[(img + (long)h * W + tw0 + t) * ldy + yoff + (c0 + cl) / 2] = s2;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < CT_TW; ++t)
        if (tw0 + t < W) y[(img + (long)h * W + tw0 + t) * ldy + yoff + c0 + cl] = acc[t];
}

template <int WIN, bool FLIP>
__global__ __launch_bounds__(256) void fa_conv_tile_kernel(const float* __restrict__ x, long ldx, int xoff,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, long ldy, int yoff,
                                                           int H, int W, int ncls, int tiles_w, int add_center) {
    constexpr int R = WIN / 2, LH = CT_TH + 2 * R, LW = CT_TW + 2 * R;
    __shared__ __attribute__((aligned(16))) float sx[LH * LW * CT_CL];
    __shared__ float sw[CT_CL * WIN * WIN];
    conv_tile_body<WIN, FLIP>(sx, sw, blockIdx.y, x, ldx, xoff, w, bias, y, ldy, yoff, H, W, ncls, tiles_w, add_center);
}

// ---- two output rows per thread on packed fp32 FMAs -----------------------------------------------------------------------------
// The stencil is VALU-bound (16 * WIN FMAs per window row and thread against ~WIN + 22 LDS reads).  Here a thread owns TWO adjacent output
// rows of its channel: window row i of the upper output row and window row i of the lower one are the LDS rows r + i and r + i + 1, read as
// ONE register pair per position (two LDS words a whole window row apart), and every tap is one v_pk_fma_f32 on (upper, lower) -- half the
// FMA instructions per output.  Tile: 16 rows x TW2 columns x 32 channels (8 row pairs x 32 channels = 256 threads); the taller tile also
// re-reads less halo (7x7: 1.9 instead of 2.4 input positions per output at 16 x 16).
typedef float ct_f2 __attribute__((ext_vector_type(2)));
constexpr int CT2_TH = 16, CT2_TW = 16;

template <int WIN, bool FLIP>
__device__ __forceinline__ void conv_tile2_body(float* __restrict__ sx, float* __restrict__ sw, int cblock,
                                                const float* __restrict__ x, long ldx, int xoff,
                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                float* __restrict__ y, long ldy, int yoff,
                                                int H, int W, int ncls, int tiles_w, int add_center) {
    constexpr int R = WIN / 2, LH = CT2_TH + 2 * R, LW = CT2_TW + 2 * R;
    const int b = blockIdx.z, c0 = cblock * CT_CL;
    const int th0 = (blockIdx.x / tiles_w) * CT2_TH, tw0 = (blockIdx.x % tiles_w) * CT2_TW;
    const int cl = threadIdx.x & 31, rp = threadIdx.x >> 5;
    const long img = (long)b * H * W;
    CtWindow<LH, LW> win;
    win.load(x + xoff + c0, ldx, img, th0 - R, tw0 - R, H, W, ncls - c0);
    constexpr int NWQ = (CT_CL * WIN * WIN + 255) / 256;
    float wq[NWQ];
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
        const int i = threadIdx.x + 256 * j, c = i / (WIN * WIN), t = i % (WIN * WIN);
        const bool ok = i < CT_CL * WIN * WIN && c0 + c < ncls;
        const float wv = w[ok ? (long)(c0 + c) * WIN * WIN + (FLIP ? WIN * WIN - 1 - t : t) : 0];
        wq[j] = ok ? wv : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NWQ; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < CT_CL * WIN * WIN) sw[i] = wq[j];
    }
    win.store(sx);
    __syncthreads();
    const int h = th0 + 2 * rp;
    if (c0 + cl >= ncls || h >= H) return;
    ct_f2 acc[CT2_TW];
    const float b0 = (!FLIP && bias) ? bias[c0 + cl] : 0.f;
#pragma unroll
    for (int t = 0; t < CT2_TW; ++t) acc[t] = ct_f2{b0, b0};
    // (one window row at a time for the wide windows: fully unrolled, hipcc hoists every row's LDS reads and needs 400 registers)
#pragma unroll(WIN > 3 ? 1 : WIN)
    for (int i = 0; i < WIN; ++i) {
        ct_f2 row[LW];
        float wr[WIN];
#pragma unroll
        for (int t = 0; t < LW; ++t) row[t] = ct_f2{sx[((2 * rp + i) * LW + t) * CT_CL + cl], sx[((2 * rp + i + 1) * LW + t) * CT_CL + cl]};
        if (i == R && add_center) {          // y = x + conv(x)  (ConvPosEnc) / dx = g + conv^T(g): both rows' centre positions sit in window row R of the pair
#pragma unroll
            for (int t = 0; t < CT2_TW; ++t) acc[t] += row[t + R];
        }
#pragma unroll
        for (int j = 0; j < WIN; ++j) wr[j] = sw[cl * WIN * WIN + i * WIN + j];
#pragma unroll
        for (int t = 0; t < CT2_TW; ++t)
#pragma unroll
            for (int j = 0; j < WIN; ++j) acc[t] = __builtin_elementwise_fma(ct_f2{wr[j], wr[j]}, row[t + j], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < CT2_TW; ++t)
        if (tw0 + t < W) y[(img + (long)h * W + tw0 + t) * ldy + yoff + c0 + cl] = acc[t].x;
    if (h + 1 < H) {
#pragma unroll
        for (int t = 0; t < CT2_TW; ++t)
            if (tw0 + t < W) y[(img + (long)(h + 1) * W + tw0 + t) * ldy + yoff + c0 + cl] = acc[t].y;
    }
}

template <int WIN, bool FLIP>
__global__ __launch_bounds__(256) void fa_conv_tile2_kernel(const float* __restrict__ x, long ldx, int xoff,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, long ldy, int yoff,
                                                            int H, int W, int ncls, int tiles_w, int add_center) {
    constexpr int R = WIN / 2, LH = CT2_TH + 2 * R, LW = CT2_TW + 2 * R;
    extern __shared__ __attribute__((aligned(16))) float ct2_smem[];
    float* sx = ct2_smem;
    float* sw = sx + LH * LW * CT_CL;
    conv_tile2_body<WIN, FLIP>(sx, sw, blockIdx.y, x, ldx, xoff, w, bias, y, ldy, yoff, H, W, ncls, tiles_w, add_center);
}

// The three window classes of ConvRelPosEnc (3x3 / 5x5 / 7x7 head groups, mpvit.py:296-318) in ONE launch: blockIdx.y walks the
// 32-channel blocks of class 3, then 5, then 7 (a third of the launches of the attention's forward and data-gradient passes).
struct Conv3Args {
    const float* x; long ldx; int xoff[3];
    const float* w[3]; const float* bias[3];
    float* y; long ldy; int yoff[3];
    int H, W, ncls[3], yb[3], tiles_w;
};
template <bool FLIP>
__global__ __launch_bounds__(256) void fa_conv3_kernel(Conv3Args p) {
    __shared__ __attribute__((aligned(16))) float sx[(CT_TH + 6) * (CT_TW + 6) * CT_CL];
    __shared__ float sw[CT_CL * 49];
    int yb = blockIdx.y;
    if (yb < p.yb[0]) { conv_tile_body<3, FLIP>(sx, sw, yb, p.x, p.ldx, p.xoff[0], p.w[0], p.bias[0], p.y, p.ldy, p.yoff[0], p.H, p.W, p.ncls[0], p.tiles_w, 0); return; }
    yb -= p.yb[0];
    if (yb < p.yb[1]) { conv_tile_body<5, FLIP>(sx, sw, yb, p.x, p.ldx, p.xoff[1], p.w[1], p.bias[1], p.y, p.ldy, p.yoff[1], p.H, p.W, p.ncls[1], p.tiles_w, 0); return; }
    yb -= p.yb[1];
    conv_tile_body<7, FLIP>(sx, sw, yb, p.x, p.ldx, p.xoff[2], p.w[2], p.bias[2], p.y, p.ldy, p.yoff[2], p.H, p.W, p.ncls[2], p.tiles_w, 0);
}

template <bool FLIP>
__global__ __launch_bounds__(256) void fa_conv3v2_kernel(Conv3Args p) {
    extern __shared__ __attribute__((aligned(16))) float ct2_smem[];
    float* sx = ct2_smem;
    float* sw = sx + (CT2_TH + 6) * (CT2_TW + 6) * CT_CL;
    int yb = blockIdx.y;
    if (yb < p.yb[0]) { conv_tile2_body<3, FLIP>(sx, sw, yb, p.x, p.ldx, p.xoff[0], p.w[0], p.bias[0], p.y, p.ldy, p.yoff[0], p.H, p.W, p.ncls[0], p.tiles_w, 0); return; }
    yb -= p.yb[0];
    if (yb < p.yb[1]) { conv_tile2_body<5, FLIP>(sx, sw, yb, p.x, p.ldx, p.xoff[1], p.w[1], p.bias[1], p.y, p.ldy, p.yoff[1], p.H, p.W, p.ncls[1], p.tiles_w, 0); return; }
    yb -= p.yb[1];
    conv_tile2_body<7, FLIP>(sx, sw, yb, p.x, p.ldx, p.xoff[2], p.w[2], p.bias[2], p.y, p.ldy, p.yoff[2], p.H, p.W, p.ncls[2], p.tiles_w, 0);
}

// 1: two rows per thread on packed FMAs (16 x 16 tiles); 0: one row per thread (8 x 16 tiles) -- MDVIT_CONV_TILE2=0 for A/B
inline bool conv_tile_v2() {
    static const bool on = [] { const char* e = getenv("MDVIT_CONV_TILE2"); return !(e && e[0] == '0'); }();
    return on;
}

template <bool FLIP>
void launch_conv3(const float* x, long ldx, const int xoff[3], const float* const w[3], const float* const bias[3], float* y, long ldy,
                  const int yoff[3], const CtGeom& g, const int ncls[3], hipStream_t s) {
    Conv3Args a;
    a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.H = g.H; a.W = g.W; a.tiles_w = cdiv(g.W, CT_TW);
    int total = 0;
    for (int i = 0; i < 3; ++i) {
        a.xoff[i] = xoff[i]; a.yoff[i] = yoff[i]; a.w[i] = w[i]; a.bias[i] = bias ? bias[i] : nullptr; a.ncls[i] = ncls[i];
        a.yb[i] = ncls[i] > 0 ? cdiv(ncls[i], CT_CL) : 0;
        total += a.yb[i];
    }
    if (total == 0) return;
    // measured (tools/conv_tile_check.py, block_roofline at bs=32): the packed two-row tiles win on the 128 x 128 / 64 x 64 token images of stages 0 / 1
    // (block forward 1.158 -> 1.108 ms, backward 2.419 -> 2.398 ms at stage 0) and lose a little on 32 x 32 and 16 x 16, where a 16 x 16 tile is a
    // quarter or all of the image
    if (conv_tile_v2() && (long)g.H * g.W >= 4096) {
        a.tiles_w = cdiv(g.W, CT2_TW);
        constexpr int smem = ((CT2_TH + 6) * (CT2_TW + 6) * CT_CL + CT_CL * 49) * 4;
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_conv3v2_kernel<FLIP>), hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_set = true; }
        hipLaunchKernelGGL((fa_conv3v2_kernel<FLIP>), dim3(a.tiles_w * cdiv(g.H, CT2_TH), total, g.B), dim3(256), smem, s, a);
        return;
    }
    hipLaunchKernelGGL((fa_conv3_kernel<FLIP>), dim3(a.tiles_w * cdiv(g.H, CT_TH), total, g.B), dim3(256), 0, s, a);
}

// dw[ci][i][j] = sum_tokens g[n,cg] * x[n + (i-R, j-R), cx];  db[ci] = sum g.   Thread (channel, window row i)
// slides along w with the x row in registers; every (channel, tap) of the block's 32 channels is owned by one thread,
// which writes its partial into the block's row of `part` ([y][z*x][32*(WIN*WIN+1)]); mdvit_reduce_partials adds the
// rows in a fixed order (deterministic; ~1000 same-address float atomics per tap cost more than the whole kernel).
// GDIV = 2: channel c of x pairs with gradient channel c / 2 (the grouped decoder conv on (skip, up): output group g reads inputs 2g, 2g+1)
// (bx, gx: this workgroup's tile chunk and the number of chunks; cb: 32-channel block inside the class; b, nb: image and image count)
template <int WIN, int GDIV>
__device__ __forceinline__ void conv_tile_wgrad_body(float* __restrict__ sx, float* __restrict__ sg, int bx, int gx, int cb, int b, int nb,
                                                     const float* __restrict__ g, long ldg, int goff,
                                                     const float* __restrict__ x, long ldx, int xoff,
                                                     float* __restrict__ part,
                                                     int H, int W, int ncls, int tiles_w, int tiles_total, int tiles_per_block) {
    constexpr int R = WIN / 2, LH = CT_TH + 2 * R, LW = CT_TW + 2 * R;
    const int c0 = cb * CT_CL;
    const int cl = threadIdx.x & 31, rl8 = threadIdx.x >> 5;
    // thread <-> (channel, window row i [, half of the tile's rows]): WIN = 3 would leave five of the eight row lanes idle, so there two lanes
    // share a window row, each over four of the tile's eight rows, and are added through LDS at the end; lanes past WIN * HS only help loading
    constexpr int HS = WIN == 3 ? 2 : 1, RPH = CT_TH / HS;
    const int rl = HS == 2 ? (rl8 < 6 ? rl8 % 3 : WIN) : rl8, half = HS == 2 ? rl8 / 3 : 0;
    const long img = (long)b * H * W;
    const bool chan_ok = c0 + cl < ncls;
    float acc[WIN];
    float accb = 0.f;
#pragma unroll
    for (int j = 0; j < WIN; ++j) acc[j] = 0.f;
    const int t_beg = bx * tiles_per_block, t_end = min(tiles_total, t_beg + tiles_per_block);
    for (int tile = t_beg; tile < t_end; ++tile) {
        const int th0 = (tile / tiles_w) * CT_TH, tw0 = (tile % tiles_w) * CT_TW;
        {   // both windows requested before either is stored
            CtWindow<LH, LW> wx;
            CtWindow<CT_TH, CT_TW> wg;
            wx.load(x + xoff + c0, ldx, img, th0 - R, tw0 - R, H, W, ncls - c0);
            wg.load(g + goff + c0 / GDIV, ldg, img, th0, tw0, H, W, (ncls - c0) / GDIV);
            wx.store(sx);
            wg.store(sg);
        }
        __syncthreads();
        if (rl < WIN && chan_ok) {
#pragma unroll 2
            for (int h = half * RPH; h < (half + 1) * RPH; ++h) {
                float row[LW], gr[CT_TW];
#pragma unroll
                for (int t = 0; t < LW; ++t) row[t] = sx[((h + rl) * LW + t) * CT_CL + cl];
#pragma unroll
                for (int t = 0; t < CT_TW; ++t) gr[t] = sg[(h * CT_TW + t) * CT_CL + cl / GDIV];
#pragma unroll
                for (int t = 0; t < CT_TW; ++t) {
                    if (rl == 0) accb += gr[t];
#pragma unroll
                    for (int j = 0; j < WIN; ++j) acc[j] = fmaf(gr[t], row[t + j], acc[j]);
                }
            }
        }
        __syncthreads();
    }
    if (HS == 2) {               // the upper row half's sums join the lower half's (sx is free: the tile loop ended with a barrier)
        float* s_h = sx;         // [3 window rows][WIN + 1][32 channels]
        if (rl < WIN && half == 1) {
#pragma unroll
            for (int j = 0; j < WIN; ++j) s_h[(rl * (WIN + 1) + j) * CT_CL + cl] = acc[j];
            s_h[(rl * (WIN + 1) + WIN) * CT_CL + cl] = accb;
        }
        __syncthreads();
        if (rl < WIN && half == 0) {
#pragma unroll
            for (int j = 0; j < WIN; ++j) acc[j] += s_h[(rl * (WIN + 1) + j) * CT_CL + cl];
            accb += s_h[(rl * (WIN + 1) + WIN) * CT_CL + cl];
        }
    }
    constexpr int ROW = CT_CL * (WIN * WIN + 1);
    float* prow = part + ((long)cb * nb * gx + (long)b * gx + bx) * ROW + cl * (WIN * WIN + 1);
    if (rl < WIN && half == 0) { // channels past ncls hold zeros
#pragma unroll
        for (int j = 0; j < WIN; ++j) prow[rl * WIN + j] = acc[j];
        if (rl == 0) prow[WIN * WIN] = accb;
    }
}

template <int WIN, int GDIV = 1>
__global__ __launch_bounds__(256) void fa_conv_tile_wgrad_kernel(const float* __restrict__ g, long ldg, int goff,
                                                                 const float* __restrict__ x, long ldx, int xoff,
                                                                 float* __restrict__ part,
                                                                 int H, int W, int ncls, int tiles_w, int tiles_total, int tiles_per_block) {
    constexpr int R = WIN / 2, LH = CT_TH + 2 * R, LW = CT_TW + 2 * R;
    __shared__ __attribute__((aligned(16))) float sx[LH * LW * CT_CL];
    __shared__ __attribute__((aligned(16))) float sg[CT_TH * CT_TW * CT_CL];
    conv_tile_wgrad_body<WIN, GDIV>(sx, sg, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z, gridDim.z, g, ldg, goff, x, ldx, xoff, part, H, W, ncls, tiles_w, tiles_total,
                                    tiles_per_block);
}

// The window-weight gradients of the three ConvRelPosEnc classes in ONE launch.  Each class reads its own channel range of the same dU / v rows (64 /
// 96 / 96 bytes of a 256-byte row at C = 64): as three launches every 128-byte line was fetched from HBM up to three times (PMC at the stage-0 shape
// of bs=32: 1.7 + 2.8 + 1.5 units of [tokens, C] for 2 units of data).  Here the class blocks of one tile chunk are neighbours in a LOGICAL workgroup
// order that the XCD remap keeps on one L2.
struct Conv3WArgs {
    const float* g; long ldg; int goff[3];
    const float* x; long ldx; int xoff[3];
    float* part[3];
    int H, W, ncls[3], cbs[3], tiles_w, tiles_total, tpb, gx, B;
};
__global__ __launch_bounds__(256) void fa_conv3_wgrad_kernel(Conv3WArgs p) {
    __shared__ __attribute__((aligned(16))) float sx[(CT_TH + 6) * (CT_TW + 6) * CT_CL];
    __shared__ __attribute__((aligned(16))) float sg[CT_TH * CT_TW * CT_CL];
    const unsigned lb = mdvit_xcd_logical_block();
    const int tcb = p.cbs[0] + p.cbs[1] + p.cbs[2];
    int cb = (int)(lb % tcb);
    const unsigned r = lb / tcb;
    const int bx = (int)(r % p.gx), b = (int)(r / p.gx);
    if (cb < p.cbs[0]) { conv_tile_wgrad_body<3, 1>(sx, sg, bx, p.gx, cb, b, p.B, p.g, p.ldg, p.goff[0], p.x, p.ldx, p.xoff[0], p.part[0], p.H, p.W, p.ncls[0], p.tiles_w, p.tiles_total, p.tpb); return; }
    cb -= p.cbs[0];
    if (cb < p.cbs[1]) { conv_tile_wgrad_body<5, 1>(sx, sg, bx, p.gx, cb, b, p.B, p.g, p.ldg, p.goff[1], p.x, p.ldx, p.xoff[1], p.part[1], p.H, p.W, p.ncls[1], p.tiles_w, p.tiles_total, p.tpb); return; }
    cb -= p.cbs[1];
    conv_tile_wgrad_body<7, 1>(sx, sg, bx, p.gx, cb, b, p.B, p.g, p.ldg, p.goff[2], p.x, p.ldx, p.xoff[2], p.part[2], p.H, p.W, p.ncls[2], p.tiles_w, p.tiles_total, p.tpb);
}

template <int WIN, bool FLIP>
void launch_conv_tile(const float* x, long ldx, int xoff, const float* w, const float* bias, float* y, long ldy, int yoff,
                      const CtGeom& g, int ncls, hipStream_t s, int add_center = 0) {
    if (ncls <= 0) return;
    if (WIN > 3 && conv_tile_v2() && (long)g.H * g.W >= 4096) {        // (3x3: 9 taps per output do not repay the taller tile's staging -- measured neutral to slower)
        constexpr int R = WIN / 2;
        constexpr int smem = ((CT2_TH + 2 * R) * (CT2_TW + 2 * R) * CT_CL + CT_CL * WIN * WIN) * 4;
        const int tw2 = cdiv(g.W, CT2_TW), th2 = cdiv(g.H, CT2_TH);
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fa_conv_tile2_kernel<WIN, FLIP>), hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr_set = true; }
        hipLaunchKernelGGL((fa_conv_tile2_kernel<WIN, FLIP>), dim3(tw2 * th2, cdiv(ncls, CT_CL), g.B), dim3(256), smem, s,
                           x, ldx, xoff, w, bias, y, ldy, yoff, g.H, g.W, ncls, tw2, add_center);
        return;
    }
    const int tiles_w = cdiv(g.W, CT_TW), tiles_h = cdiv(g.H, CT_TH);
    hipLaunchKernelGGL((fa_conv_tile_kernel<WIN, FLIP>), dim3(tiles_w * tiles_h, cdiv(ncls, CT_CL), g.B), dim3(256), 0, s,
                       x, ldx, xoff, w, bias, y, ldy, yoff, g.H, g.W, ncls, tiles_w, add_center);
}

// fixed-order sum of the partial rows of one 32-channel block: 32 columns x 32 row lanes per workgroup (1024 threads), eight independent loads in flight per thread.
// (Round 6: with 8 row lanes and a plain `sacc += base[..]` loop every thread walked rows / 8 partial rows one HBM round trip at a time -- 50 us per launch for the 7 x 7
// class at 256 rows, 1.3 ms per bs=4 step over the three classes on the weight-gradient stream.)
constexpr int CT_FIN_LANES = 32;
template <int WIN>
__global__ __launch_bounds__(1024) void fa_conv_wgrad_finish_kernel(const float* __restrict__ part, int nrows, float* __restrict__ dw,
                                                                    float* __restrict__ db, int ncls, int accumulate) {
    constexpr int T = WIN * WIN + 1, ROW = CT_CL * T;
    __shared__ float s_sum[CT_FIN_LANES][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;
    const float* base = part + (long)blockIdx.y * nrows * ROW;
    float sacc = 0.f;
    if (i < ROW) {
        int b = rl;
        for (; b + 7 * CT_FIN_LANES < nrows; b += 8 * CT_FIN_LANES) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(long)(b + u * CT_FIN_LANES) * ROW + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) sacc += v[u];
        }
        for (; b < nrows; b += CT_FIN_LANES) sacc += base[(long)b * ROW + i];
    }
    s_sum[rl][cl] = sacc;
    __syncthreads();
    if (rl == 0 && i < ROW) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < CT_FIN_LANES; ++r) t += s_sum[r][cl];
        const int c = blockIdx.y * CT_CL + i / T, tap = i % T;
        if (c < ncls) {
            float* dst = tap < WIN * WIN ? dw + (long)c * WIN * WIN + tap : (db ? db + c : nullptr);
            if (dst) *dst = accumulate ? *dst + t : t;
        }
    }
}

// tiles per block: keep >= ~512 blocks while halving the number of partial rows
void conv_wgrad_plan(const CtGeom& g, int ncls, int& tpb, long& nblk) {
    const int tiles = cdiv(g.W, CT_TW) * cdiv(g.H, CT_TH);
    tpb = 1;
    while (tpb < tiles && (long)cdiv(tiles, tpb * 2) * cdiv(ncls, CT_CL) * g.B >= 512) tpb *= 2;
    nblk = (long)cdiv(tiles, tpb) * cdiv(ncls, CT_CL) * g.B;
}

template <int WIN, int GDIV = 1>
int launch_conv_tile_wgrad(const float* gsrc, long ldg, int goff, const float* x, long ldx, int xoff, float* dw, float* db, float* part,
                           const CtGeom& g, int ncls, hipStream_t s, int accumulate = 0) {
    if (ncls <= 0) return MDVIT_OK;
    const int tiles_w = cdiv(g.W, CT_TW), tiles_h = cdiv(g.H, CT_TH), tiles = tiles_w * tiles_h;
    int tpb; long nblk;
    conv_wgrad_plan(g, ncls, tpb, nblk);
    const int gx = cdiv(tiles, tpb), gy = cdiv(ncls, CT_CL);
    hipLaunchKernelGGL((fa_conv_tile_wgrad_kernel<WIN, GDIV>), dim3(gx, gy, g.B), dim3(256), 0, s,
                       gsrc, ldg, goff, x, ldx, xoff, part, g.H, g.W, ncls, tiles_w, tiles, tpb);
    // second stage, per 32-channel block y: rows [y][gx*B] of 32*(WIN^2+1) floats -> dw [c][WIN^2], db [c]
    constexpr int T = WIN * WIN + 1, ROW = CT_CL * T;
    hipLaunchKernelGGL((fa_conv_wgrad_finish_kernel<WIN>), dim3(cdiv(ROW, 32), gy), dim3(32 * CT_FIN_LANES), 0, s, part, gx * g.B, dw, db, ncls, accumulate);
    return MDVIT_OK;
}

// the three classes' weight gradients: one tile launch (see fa_conv3_wgrad_kernel) + a finish per class; part: >= conv3_wgrad_part_floats floats
inline long conv3_wgrad_rows(const CtGeom& g, const int ncls[3], int& tpb, int& gx) {
    const int tiles = cdiv(g.W, CT_TW) * cdiv(g.H, CT_TH);
    const int tcb = cdiv(ncls[0], CT_CL) + cdiv(ncls[1], CT_CL) + cdiv(ncls[2], CT_CL);
    tpb = 1;
    while (tpb < tiles && (long)cdiv(tiles, tpb * 2) * tcb * g.B >= 512) tpb *= 2;
    gx = cdiv(tiles, tpb);
    return (long)gx * g.B;           // partial rows per 32-channel block
}
inline int launch_conv3_wgrad(const float* gsrc, long ldg, const int goff[3], const float* x, long ldx, const int xoff[3], float* const dw[3], float* const db[3],
                              float* part, const CtGeom& g, const int ncls[3], hipStream_t s, int accumulate) {
    int tpb, gx;
    const long rows = conv3_wgrad_rows(g, ncls, tpb, gx);
    Conv3WArgs a;
    a.g = gsrc; a.ldg = ldg; a.x = x; a.ldx = ldx; a.H = g.H; a.W = g.W; a.tiles_w = cdiv(g.W, CT_TW); a.tiles_total = a.tiles_w * cdiv(g.H, CT_TH);
    a.tpb = tpb; a.gx = gx; a.B = g.B;
    const int T[3] = {10, 26, 50};
    float* pp = part;
    int tcb = 0;
    for (int i = 0; i < 3; ++i) {
        a.goff[i] = goff[i]; a.xoff[i] = xoff[i]; a.ncls[i] = ncls[i]; a.cbs[i] = ncls[i] > 0 ? cdiv(ncls[i], CT_CL) : 0;
        a.part[i] = pp;
        pp += (long)a.cbs[i] * rows * CT_CL * T[i];
        tcb += a.cbs[i];
    }
    if (tcb == 0) return MDVIT_OK;
    hipLaunchKernelGGL(fa_conv3_wgrad_kernel, dim3((unsigned)((long)g.B * gx * tcb)), dim3(256), 0, s, a);
    if (a.cbs[0]) hipLaunchKernelGGL((fa_conv_wgrad_finish_kernel<3>), dim3(cdiv(CT_CL * 10, 32), a.cbs[0]), dim3(32 * CT_FIN_LANES), 0, s, a.part[0], (int)rows, dw[0], db[0], ncls[0], accumulate);
    if (a.cbs[1]) hipLaunchKernelGGL((fa_conv_wgrad_finish_kernel<5>), dim3(cdiv(CT_CL * 26, 32), a.cbs[1]), dim3(32 * CT_FIN_LANES), 0, s, a.part[1], (int)rows, dw[1], db[1], ncls[1], accumulate);
    if (a.cbs[2]) hipLaunchKernelGGL((fa_conv_wgrad_finish_kernel<7>), dim3(cdiv(CT_CL * 50, 32), a.cbs[2]), dim3(32 * CT_FIN_LANES), 0, s, a.part[2], (int)rows, dw[2], db[2], ncls[2], accumulate);
    return MDVIT_OK;
}

}  // namespace
