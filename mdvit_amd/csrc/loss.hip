// Step losses on logits, fused: sigmoid + BCE (nn.BCELoss: mean, log clamped at -100) + Dice
// (Utils/losses.py:8-16) for the main and auxiliary outputs + the mutual "KT" Dice between them
// (multi_train_MDViT.py:147-169).  One streaming pass produces the 8 sums; backward is one more pass.
#include "common.h"

namespace {

// sums: 0 bce_o  1 o*y  2 o*o  3 y*y  4 bce_a  5 a*y  6 a*a  7 a*o
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float bce_term(float p, float y) {
    const float lp = fmaxf(logf(p), -100.0f), l1p = fmaxf(logf(1.0f - p), -100.0f);
    return -(y * lp + (1.0f - y) * l1p);
}

__global__ __launch_bounds__(256) void seg_losses_sums_kernel(const float* __restrict__ out, const float* __restrict__ aux,
                                                              const float* __restrict__ label, double* __restrict__ sums, long n) {
    // blockIdx.y = domain batch of a domain-batched forward: n elements each, 16 sums each
    out += (long)blockIdx.y * n; label += (long)blockIdx.y * n; sums += 16 * blockIdx.y;
    if (aux) aux += (long)blockIdx.y * n;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float y = label[i], o = sigmoid_f(out[i]);
        acc[0] += bce_term(o, y); acc[1] += o * y; acc[2] += o * o; acc[3] += y * y;
        if (aux) {
            const float a = sigmoid_f(aux[i]);
            acc[4] += bce_term(a, y); acc[5] += a * y; acc[6] += a * a; acc[7] += a * o;
        }
    }
    __shared__ float s_red[4][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float v = wave_sum(acc[k]); if (lane == 0) s_red[wave][k] = v; }
    __syncthreads();
    if (threadIdx.x < 8) atomicAdd(&sums[threadIdx.x], (double)(s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]));
}

__global__ void seg_losses_final_kernel(const double* __restrict__ sums, float* __restrict__ losses, long n, int has_aux) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double eps = 1e-5, N = (double)n;
        losses[0] = (float)(sums[0] / N + 1.0 - (2.0 * sums[1] + eps) / (sums[2] + sums[3] + eps));
        if (has_aux) {
            losses[1] = (float)(sums[4] / N + 1.0 - (2.0 * sums[5] + eps) / (sums[6] + sums[3] + eps));
            losses[2] = (float)(1.0 - (2.0 * sums[7] + eps) / (sums[6] + sums[2] + eps));
        } else { losses[1] = 0.f; losses[2] = 0.f; }
    }
}

// G domain batches: per-batch losses (per_group [G][3], optional) and their sums in batch order -- ((l_0 + l_1) + l_2) + ... in fp32, what the step's own
// additions of the per-domain losses produced
__global__ void seg_losses_groups_final_kernel(const double* __restrict__ sums, float* __restrict__ losses, float* __restrict__ per_group, long n, int has_aux, int G) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double eps = 1e-5, N = (double)n;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int g = 0; g < G; ++g) {
            const double* s = sums + 16 * g;
            const float l0 = (float)(s[0] / N + 1.0 - (2.0 * s[1] + eps) / (s[2] + s[3] + eps));
            float l1 = 0.f, l2 = 0.f;
            if (has_aux) {
                l1 = (float)(s[4] / N + 1.0 - (2.0 * s[5] + eps) / (s[6] + s[3] + eps));
                l2 = (float)(1.0 - (2.0 * s[7] + eps) / (s[6] + s[2] + eps));
            }
            if (per_group) { per_group[3 * g] = l0; per_group[3 * g + 1] = l1; per_group[3 * g + 2] = l2; }
            t0 = g ? t0 + l0 : l0; t1 = g ? t1 + l1 : l1; t2 = g ? t2 + l2 : l2;
        }
        losses[0] = t0; losses[1] = t1; losses[2] = t2;
    }
}

// d/ds Dice(s,t) = -(2 t D - (2I+eps) 2 s) / D^2,  D = sum s^2 + sum t^2 + eps
__global__ __launch_bounds__(256) void seg_losses_bwd_kernel(const float* __restrict__ out, const float* __restrict__ aux,
                                                             const float* __restrict__ label, const double* __restrict__ sums,
                                                             const float* __restrict__ gp0, const float* __restrict__ gp1, const float* __restrict__ gp2,
                                                             float* __restrict__ dout, float* __restrict__ daux, long n, float dice_gain) {
    out += (long)blockIdx.y * n; label += (long)blockIdx.y * n; sums += 16 * blockIdx.y;      // blockIdx.y = domain batch (see the sums kernel)
    if (aux) aux += (long)blockIdx.y * n;
    if (dout) dout += (long)blockIdx.y * n;
    if (daux) daux += (long)blockIdx.y * n;
    const float eps = 1e-5f, invN = 1.0f / (float)n;
    const float g0 = gp0 ? gp0[0] : 0.f, g1 = gp1 ? gp1[0] : 0.f, g2 = (gp2 ? gp2[0] : 0.f) * dice_gain;      // a NULL upstream gradient: that loss takes no part in this sweep
    const float D_o = (float)(sums[2] + sums[3]) + eps, I_o = 2.f * (float)sums[1] + eps;
    const float D_a = (float)(sums[6] + sums[3]) + eps, I_a = 2.f * (float)sums[5] + eps;
    const float D_k = (float)(sums[6] + sums[2]) + eps, I_k = 2.f * (float)sums[7] + eps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float y = label[i], o = sigmoid_f(out[i]);
        const float so = o * (1.f - o);
        float go = 0.f, a = 0.f;
        if (aux) a = sigmoid_f(aux[i]);
        if (dout) {
            const float dbce = (o - y) / fmaxf((1.f - o) * o, 1e-12f) * invN;
            const float ddice = -(2.f * y * D_o - I_o * 2.f * o) / (D_o * D_o) * dice_gain;
            go = g0 * (dbce + ddice);
            if (aux) go += g2 * (-(2.f * a * D_k - I_k * 2.f * o) / (D_k * D_k));
            dout[i] = go * so;
        }
        if (aux && daux) {
            const float sa = a * (1.f - a);
            const float dbce = (a - y) / fmaxf((1.f - a) * a, 1e-12f) * invN;
            const float ddice = -(2.f * y * D_a - I_a * 2.f * a) / (D_a * D_a) * dice_gain;
            const float dkt = -(2.f * o * D_k - I_k * 2.f * a) / (D_k * D_k);
            daux[i] = (g1 * (dbce + ddice) + g2 * dkt) * sa;
        }
    }
}

}  // namespace

// ---- on-device segmentation metrics (multi_train_MDViT.py:172-179,275-288 with medpy.metric.binary dc / jc) ------------
// counts (int64): 0 |A&Y|  1 |A|  2 |Y|  3 |Aaux&Y|  4 |Aaux|,  A = sigmoid(out) > 0.5, Y = label != 0
namespace {
__global__ __launch_bounds__(256) void seg_metric_counts_kernel(const float* __restrict__ out, const float* __restrict__ aux,
                                                                const float* __restrict__ label, unsigned long long* __restrict__ counts, long n) {
    unsigned int c[5] = {0, 0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const bool y = label[i] != 0.f, a = sigmoid_f(out[i]) > 0.5f;
        c[0] += a && y; c[1] += a; c[2] += y;
        if (aux) { const bool b = sigmoid_f(aux[i]) > 0.5f; c[3] += b && y; c[4] += b; }
    }
    __shared__ unsigned int s_c[4][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        unsigned int v = c[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_c[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(&counts[threadIdx.x], (unsigned long long)(s_c[0][threadIdx.x] + s_c[1][threadIdx.x] + s_c[2][threadIdx.x] + s_c[3][threadIdx.x]));
}

// metrics: 0 dice(out)  1 iou(out)  2 dice(aux)  3 iou(aux);  0/0 -> 0 (medpy's dc returns 0.0 there; its jc would raise)
__global__ void seg_metric_final_kernel(const unsigned long long* __restrict__ counts, float* __restrict__ metrics, int has_aux) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double i0 = (double)counts[0], a0 = (double)counts[1], y = (double)counts[2];
        metrics[0] = (a0 + y) > 0 ? (float)(2.0 * i0 / (a0 + y)) : 0.f;
        metrics[1] = (a0 + y - i0) > 0 ? (float)(i0 / (a0 + y - i0)) : 0.f;
        const double i1 = (double)counts[3], a1 = (double)counts[4];
        metrics[2] = has_aux && (a1 + y) > 0 ? (float)(2.0 * i1 / (a1 + y)) : 0.f;
        metrics[3] = has_aux && (a1 + y - i1) > 0 ? (float)(i1 / (a1 + y - i1)) : 0.f;
    }
}

// ---- input pipeline: uint8 HWC image -> ImageNet-normalised fp32 CHW (create_dataset.py:25-26,143-144,165-172) --------
// norm01 divides in float64 and the result is cast to float32; Normalize then subtracts the mean and divides by the std in fp32
__global__ __launch_bounds__(256) void image_normalize_u8_kernel(const unsigned char* __restrict__ img, float* __restrict__ out, int B, int H, int W) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const long npix = (long)B * H * W;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const long b = p / ((long)H * W), hw = p % ((long)H * W);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)((double)img[p * 3 + c] / 255.0);
            out[(b * 3 + c) * (long)H * W + hw] = (v - mean[c]) / stdv[c];
        }
    }
}
}  // namespace

extern "C" int mdvit_seg_metrics(const float* out, const float* aux, const float* label, uint64_t* counts, float* metrics, int64_t n, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(out && label && counts && metrics && n > 0, MDVIT_E_SHAPE, "seg_metrics: bad arguments");
    MDVIT_ZERO(counts, sizeof(uint64_t) * 8, s);
    const int grid = (int)((n + 256L * 8 - 1) / (256L * 8) < 1024 ? (n + 256L * 8 - 1) / (256L * 8) : 1024);
    hipLaunchKernelGGL(seg_metric_counts_kernel, dim3(grid), dim3(256), 0, s, out, aux, label, (unsigned long long*)counts, (long)n);
    hipLaunchKernelGGL(seg_metric_final_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long*)counts, metrics, aux != nullptr);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_image_normalize_u8(const uint8_t* img_nhwc, float* out_nchw, int32_t B, int32_t H, int32_t W, void* stream) {
    MDVIT_CHECK_ARG(img_nhwc && out_nchw && B > 0 && H > 0 && W > 0, MDVIT_E_SHAPE, "image_normalize_u8: bad arguments");
    const long npix = (long)B * H * W;
    hipLaunchKernelGGL(image_normalize_u8_kernel, dim3((int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192)), dim3(256), 0, (hipStream_t)stream,
                       img_nhwc, out_nchw, B, H, W);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

static int seg_sums(const float* out, const float* aux, const float* label, double* sums, int64_t n, hipStream_t s) {
    MDVIT_ZERO(sums, sizeof(double) * 16, s);
    const int grid = (int)((n + 256L * 8 - 1) / (256L * 8) < 1024 ? (n + 256L * 8 - 1) / (256L * 8) : 1024);
    hipLaunchKernelGGL(seg_losses_sums_kernel, dim3(grid), dim3(256), 0, s, out, aux, label, sums, (long)n);
    return MDVIT_OK;
}

extern "C" int mdvit_seg_losses_sums(const float* out, const float* aux, const float* label, double* sums, int64_t n, void* stream) {
    MDVIT_CHECK_ARG(out && label && sums && n > 0, MDVIT_E_SHAPE, "seg_losses_sums: bad arguments");
    const int rc = seg_sums(out, aux, label, sums, n, (hipStream_t)stream);
    if (rc != MDVIT_OK) return rc;
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_seg_losses_final(const double* sums, float* losses, int64_t n_total, int32_t has_aux, void* stream) {
    MDVIT_CHECK_ARG(sums && losses && n_total > 0, MDVIT_E_SHAPE, "seg_losses_final: bad arguments");
    hipLaunchKernelGGL(seg_losses_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, losses, (long)n_total, has_aux);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_seg_losses_fwd(const float* out, const float* aux, const float* label, double* sums, float* losses, int64_t n, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(out && label && sums && losses && n > 0, MDVIT_E_SHAPE, "seg_losses_fwd: bad arguments");
    const int rc = seg_sums(out, aux, label, sums, n, s);
    if (rc != MDVIT_OK) return rc;
    hipLaunchKernelGGL(seg_losses_final_kernel, dim3(1), dim3(64), 0, s, sums, losses, (long)n, aux != nullptr);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_seg_losses_bwd(const float* out, const float* aux, const float* label, const double* sums, const float* g,
                                    float* dout, float* daux, int64_t n, float dice_gain, void* stream) {
    MDVIT_CHECK_ARG(out && label && sums && g && n > 0, MDVIT_E_SHAPE, "seg_losses_bwd: bad arguments");
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(seg_losses_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, aux, label, sums, g, g + 1, g + 2, dout, daux, (long)n, dice_gain);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_seg_losses_bwd3(const float* out, const float* aux, const float* label, const double* sums, const float* g0, const float* g1, const float* g2,
                                     float* dout, float* daux, int64_t n, float dice_gain, void* stream) {
    MDVIT_CHECK_ARG(out && label && sums && (g0 || g1 || g2) && n > 0, MDVIT_E_SHAPE, "seg_losses_bwd3: bad arguments");
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(seg_losses_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, aux, label, sums, g0, g1, g2, dout, daux, (long)n, dice_gain);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

/* The G domain batches of a domain-batched forward in one launch each way: out / aux / label hold G consecutive batches of n elements, sums [G][16]. */
extern "C" int mdvit_seg_losses_groups_sums(const float* out, const float* aux, const float* label, double* sums, int64_t n, int32_t G, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(out && label && sums && n > 0 && G > 0 && G <= 64, MDVIT_E_SHAPE, "seg_losses_groups_sums: bad arguments (n=%ld G=%d)", (long)n, G);
    MDVIT_ZERO(sums, sizeof(double) * 16 * G, s);
    const int grid = (int)((n + 256L * 8 - 1) / (256L * 8) < 1024 ? (n + 256L * 8 - 1) / (256L * 8) : 1024);
    hipLaunchKernelGGL(seg_losses_sums_kernel, dim3(grid, G), dim3(256), 0, s, out, aux, label, sums, (long)n);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_seg_losses_groups_final(const double* sums, float* losses, float* per_group, int64_t n_total, int32_t has_aux, int32_t G, void* stream) {
    MDVIT_CHECK_ARG(sums && losses && n_total > 0 && G > 0 && G <= 64, MDVIT_E_SHAPE, "seg_losses_groups_final: bad arguments");
    hipLaunchKernelGGL(seg_losses_groups_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, losses, per_group, (long)n_total, has_aux, G);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_seg_losses_groups_bwd(const float* out, const float* aux, const float* label, const double* sums, const float* g0, const float* g1, const float* g2,
                                           float* dout, float* daux, int64_t n, int32_t G, float dice_gain, void* stream) {
    MDVIT_CHECK_ARG(out && label && sums && (g0 || g1 || g2) && n > 0 && G > 0 && G <= 64, MDVIT_E_SHAPE, "seg_losses_groups_bwd: bad arguments");
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(seg_losses_bwd_kernel, dim3(grid, G), dim3(256), 0, (hipStream_t)stream, out, aux, label, sums, g0, g1, g2, dout, daux, (long)n, dice_gain);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
