// AdamW over the whole parameter set in ONE launch (optim.AdamW of multi_train_MDViT.py:91-93).
// The parameters are separate tensors, their gradients views into flat buckets (parallel.GradAccumulator), the moments flat
// buffers with the bucket layout: a device table of {param, grad, exp_avg, exp_avg_sq, numel} rows drives the kernel
// (grid.y = table row).  Step count and learning rate live in device memory, so a captured HIP graph replays it unchanged.
#include "common.h"

namespace {

struct AdamScalars { float beta1, beta2, eps, weight_decay; int zero_grad; };

// torch.optim.AdamW (amsgrad=False, maximize=False), single-tensor formulation:
//   p *= 1 - lr*wd;  m = lerp(m, g, 1-b1);  v = b2*v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(const long long* __restrict__ table, const float* __restrict__ lr_dev,
                                                    const float* __restrict__ step_dev, AdamScalars h) {
    const long long* row = table + 5 * (long)blockIdx.y;
    float* p = reinterpret_cast<float*>(row[0]);
    float* g = reinterpret_cast<float*>(row[1]);
    float* m = reinterpret_cast<float*>(row[2]);
    float* v = reinterpret_cast<float*>(row[3]);
    const long n = (long)row[4];
    const float lr = lr_dev[0], t = step_dev[0];          // t already counts this step (adamw_tick_kernel ran first)
    const float bc1 = 1.0f - powf(h.beta1, t), bc2 = 1.0f - powf(h.beta2, t);
    const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2), decay = 1.0f - lr * h.weight_decay;
    const long nq = n >> 2;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        for (long q = t0; q < nq; q += stride) {
            float4 pv = reinterpret_cast<float4*>(p)[q], gv = reinterpret_cast<float4*>(g)[q];
            float4 mv = reinterpret_cast<float4*>(m)[q], vv = reinterpret_cast<float4*>(v)[q];
            float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w}, vq[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pp[j] *= decay;
                mm[j] = mm[j] + (gg[j] - mm[j]) * (1.0f - h.beta1);
                vq[j] = h.beta2 * vq[j] + (1.0f - h.beta2) * gg[j] * gg[j];
                pp[j] -= step_size * mm[j] / (sqrtf(vq[j]) * inv_sqrt_bc2 + h.eps);
            }
            reinterpret_cast<float4*>(p)[q] = make_float4(pp[0], pp[1], pp[2], pp[3]);
            reinterpret_cast<float4*>(m)[q] = make_float4(mm[0], mm[1], mm[2], mm[3]);
            reinterpret_cast<float4*>(v)[q] = make_float4(vq[0], vq[1], vq[2], vq[3]);
            if (h.zero_grad) reinterpret_cast<float4*>(g)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (long i = (vec ? nq * 4 : 0) + t0; i < n; i += stride) {
        float pp = p[i] * decay;
        const float gg = g[i];
        const float mm = m[i] + (gg - m[i]) * (1.0f - h.beta1);
        const float vq = h.beta2 * v[i] + (1.0f - h.beta2) * gg * gg;
        pp -= step_size * mm / (sqrtf(vq) * inv_sqrt_bc2 + h.eps);
        p[i] = pp; m[i] = mm; v[i] = vq;
        if (h.zero_grad) g[i] = 0.f;
    }
}

__global__ void adamw_tick_kernel(float* step_dev) { if (threadIdx.x == 0 && blockIdx.x == 0) step_dev[0] += 1.0f; }

}  // namespace

extern "C" int mdvit_adamw_step(const void* table_dev, int32_t n_tensors, int32_t blocks_per_tensor, const float* lr_dev, float* step_dev,
                                float beta1, float beta2, float eps, float weight_decay, int32_t zero_grad, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(table_dev && lr_dev && step_dev && n_tensors > 0 && blocks_per_tensor > 0, MDVIT_E_SHAPE, "adamw_step: bad arguments");
    MDVIT_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, MDVIT_E_SHAPE, "adamw_step: bad hyper-parameters");
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(64), 0, s, step_dev);
    AdamScalars h{beta1, beta2, eps, weight_decay, zero_grad};
    // gridDim.y is limited to 65535: a table with more rows (one row per 32768-element chunk: > 2.1 G parameters) goes out in slices of the table
    for (int32_t r0 = 0; r0 < n_tensors; r0 += 65535) {
        const int32_t nr = n_tensors - r0 < 65535 ? n_tensors - r0 : 65535;
        hipLaunchKernelGGL(adamw_kernel, dim3(blocks_per_tensor, nr), dim3(256), 0, s, (const long long*)table_dev + 5 * (long long)r0, lr_dev, (const float*)step_dev, h);
        MDVIT_LAUNCH_CHECK();
    }
    return MDVIT_OK;
}
