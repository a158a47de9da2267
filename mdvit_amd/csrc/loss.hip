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

// d/ds Dice(s,t) = -(2 t D - (2I+eps) 2 s) / D^2,  D = sum s^2 + sum t^2 + eps
__global__ __launch_bounds__(256) void seg_losses_bwd_kernel(const float* __restrict__ out, const float* __restrict__ aux,
                                                             const float* __restrict__ label, const double* __restrict__ sums,
                                                             const float* __restrict__ g, float* __restrict__ dout, float* __restrict__ daux, long n,
                                                             float dice_gain) {
    const float eps = 1e-5f, invN = 1.0f / (float)n;
    const float g0 = g[0], g1 = g[1], g2 = g[2] * dice_gain;
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
    hipLaunchKernelGGL(seg_losses_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, aux, label, sums, g, dout, daux, (long)n, dice_gain);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
