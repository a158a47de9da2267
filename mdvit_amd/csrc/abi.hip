// Error plumbing and version of libmdvit_hip.so.
#include <stdarg.h>

#include "common.h"

thread_local char g_mdvit_err[512] = {0};

int mdvit_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_mdvit_err, sizeof(g_mdvit_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* mdvit_last_error(void) { return g_mdvit_err; }
extern "C" int mdvit_version(void) { return MDVIT_ABI_VERSION; }

namespace {
__global__ __launch_bounds__(256) void zero_u32_kernel(uint32_t* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0u;
}
__global__ __launch_bounds__(256) void zero_u128_kernel(uint4* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
}  // namespace

int mdvit_zero_async(void* ptr, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return MDVIT_OK;
    if (ptr == nullptr || (bytes & 3) || (reinterpret_cast<uintptr_t>(ptr) & 3))
        return mdvit_set_error(MDVIT_E_ALIGN, "zero fill needs a 4-byte aligned buffer and size (ptr=%p bytes=%zu)", ptr, bytes);
    if (((reinterpret_cast<uintptr_t>(ptr) | bytes) & 15) == 0) {
        const size_t n = bytes / 16;
        const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(zero_u128_kernel, dim3(grid), dim3(256), 0, stream, (uint4*)ptr, n);
    } else {
        const size_t n = bytes / 4;
        const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(zero_u32_kernel, dim3(grid), dim3(256), 0, stream, (uint32_t*)ptr, n);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "zero fill launch failed: %s", hipGetErrorString(e));
    return MDVIT_OK;
}

namespace {
// 8 columns x 32 row lanes per workgroup: lane r adds rows r, r+32, ... (4 loads in flight), the 32 lane sums are folded by a
// fixed shuffle tree.  (Few outputs, up to ~1000 rows: the serial row walk, not bandwidth, sets the time.)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nblk, long stride, int n0,
                                                              float* __restrict__ out0, int n1, float* __restrict__ out1, int accumulate) {
    const int n = n0 + n1;
    const int rl = threadIdx.x & 31, cl = threadIdx.x >> 5;      // a 32-lane half-wavefront per column
    const int i = blockIdx.x * 8 + cl;
    part += (long)blockIdx.y * nblk * stride;        // batched mode (grid.y > 1): one reduction per batch, out0 [batch][n0]
    out0 += (long)blockIdx.y * n0;
    if (out1) out1 += (long)blockIdx.y * n1;
    float s = 0.f;
    if (i < n) {
        int b = rl;
        for (; b + 96 < nblk; b += 128) {
            const float v0 = part[(long)b * stride + i], v1 = part[(long)(b + 32) * stride + i];
            const float v2 = part[(long)(b + 64) * stride + i], v3 = part[(long)(b + 96) * stride + i];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; b < nblk; b += 32) s += part[(long)b * stride + i];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_down(s, o, 32);
    if (rl == 0 && i < n) {
        float* dst = i < n0 ? out0 + i : (out1 ? out1 + (i - n0) : nullptr);
        if (dst) *dst = accumulate ? *dst + s : s;
    }
}
}  // namespace

int mdvit_reduce_partials(const float* part, int nblk, long stride, int n0, float* out0, int n1, float* out1, int accumulate, hipStream_t stream) {
    if (n0 + n1 <= 0 || nblk <= 0) return MDVIT_OK;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n0 + n1 + 7) / 8), dim3(256), 0, stream, part, nblk, stride, n0, out0, n1, out1, accumulate);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

int mdvit_reduce_partials_batched(const float* part, int batches, int nblk, int n, float* out, hipStream_t stream) {
    if (n <= 0 || nblk <= 0 || batches <= 0) return MDVIT_OK;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 7) / 8, batches), dim3(256), 0, stream, part, nblk, (long)n, n, out, 0, (float*)nullptr, 0);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

int mdvit_reduce_partials_batched2(const float* part, int batches, int nblk, int n0, float* out0, int n1, float* out1, hipStream_t stream) {
    if (n0 + n1 <= 0 || nblk <= 0 || batches <= 0) return MDVIT_OK;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n0 + n1 + 7) / 8, batches), dim3(256), 0, stream, part, nblk, (long)(n0 + n1), n0, out0, n1, out1, 0);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" size_t mdvit_partials_ws_bytes(int32_t n_outputs) {
    return n_outputs > 0 ? sizeof(float) * (size_t)MDVIT_MAX_PARTIAL_ROWS * (size_t)n_outputs : 0;
}

int mdvit_zero_many(const MdvitZeroItem* items, int n, hipStream_t stream) {
    int i = 0;
    while (i < n) {
        if (items[i].p == nullptr || items[i].bytes == 0) { ++i; continue; }
        char* beg = (char*)items[i].p;
        size_t len = items[i].bytes;
        int j = i + 1;
        while (j < n && items[j].p != nullptr && (char*)items[j].p == beg + len) { len += items[j].bytes; ++j; }
        const int rc = mdvit_zero_async(beg, len, stream);
        if (rc != MDVIT_OK) return rc;
        i = j;
    }
    return MDVIT_OK;
}
