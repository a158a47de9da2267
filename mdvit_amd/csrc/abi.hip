// Error plumbing and version of libmdvit_hip.so.
#include <stdarg.h>

#include <stdlib.h>
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

hipEvent_t g_mdvit_t0 = nullptr, g_mdvit_t1 = nullptr;

extern "C" int mdvit_timing_arm(void* start_event, void* stop_event) {
    g_mdvit_t0 = (hipEvent_t)start_event;
    g_mdvit_t1 = start_event ? (hipEvent_t)stop_event : nullptr;
    return MDVIT_OK;
}
extern "C" int mdvit_event_create(void** out) {
    MDVIT_CHECK_ARG(out != nullptr, MDVIT_E_SHAPE, "event_create: null output");
    hipEvent_t e = nullptr;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "hipEventCreate failed: %s", hipGetErrorString(rc));
    *out = (void*)e;
    return MDVIT_OK;
}
extern "C" int mdvit_event_destroy(void* ev) {
    if (ev) hipEventDestroy((hipEvent_t)ev);
    return MDVIT_OK;
}
extern "C" int mdvit_event_elapsed_ms(void* start_event, void* stop_event, float* ms) {
    MDVIT_CHECK_ARG(start_event && stop_event && ms, MDVIT_E_SHAPE, "event_elapsed_ms: null argument");
    hipError_t rc = hipEventElapsedTime(ms, (hipEvent_t)start_event, (hipEvent_t)stop_event);
    if (rc != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "hipEventElapsedTime failed: %s", hipGetErrorString(rc));
    return MDVIT_OK;
}
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
// Column sums of partial rows: part [nblk][stride] -> out [n0 | n1].  A column QUAD (float4) is summed by LPQ row lanes
// (lane r adds rows r, r+LPQ, ... with all of its loads in flight at once), 256/LPQ quads per workgroup; the lane sums are
// folded by a fixed shuffle tree (and, for LPQ = 256, a fixed-order pass through LDS), so the result is deterministic.
// LPQ follows the row count: ~1000 rows x few columns wants every thread on rows (the walk is latency, not bandwidth,
// bound); a few rows x many columns (the attention's tile partials) wants the threads on columns.
template <int LPQ>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nblk, long stride, int n0,
                                                              float* __restrict__ out0, int n1, float* __restrict__ out1, int accumulate, int vec) {
    __shared__ float4 s_red[4];
    const int n = n0 + n1;
    const int rl = threadIdx.x % LPQ;
    const int c = (blockIdx.x * (256 / LPQ) + threadIdx.x / LPQ) * 4;
    part += (long)blockIdx.y * nblk * stride;        // batched mode (grid.y > 1): one reduction per batch, out [batch][n0] (+ [batch][n1])
    out0 += (long)blockIdx.y * n0;
    if (out1) out1 += (long)blockIdx.y * n1;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < n) {
        if (vec && c + 3 < n) {
            int b = rl;
            for (; b + 3 * LPQ < nblk; b += 4 * LPQ) {
                const float4 v0 = *reinterpret_cast<const float4*>(part + (long)b * stride + c);
                const float4 v1 = *reinterpret_cast<const float4*>(part + (long)(b + LPQ) * stride + c);
                const float4 v2 = *reinterpret_cast<const float4*>(part + (long)(b + 2 * LPQ) * stride + c);
                const float4 v3 = *reinterpret_cast<const float4*>(part + (long)(b + 3 * LPQ) * stride + c);
                s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
                s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
                s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
                s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
            }
            for (; b < nblk; b += LPQ) {
                const float4 v = *reinterpret_cast<const float4*>(part + (long)b * stride + c);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        } else {
            for (int b = rl; b < nblk; b += LPQ) {
                const float* r = part + (long)b * stride + c;
                s.x += r[0];
                if (c + 1 < n) s.y += r[1];
                if (c + 2 < n) s.z += r[2];
                if (c + 3 < n) s.w += r[3];
            }
        }
    }
    constexpr int W = LPQ < 64 ? LPQ : 64;
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) {
        s.x += __shfl_down(s.x, o, W); s.y += __shfl_down(s.y, o, W); s.z += __shfl_down(s.z, o, W); s.w += __shfl_down(s.w, o, W);
    }
    if (LPQ == 256) {
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float4 a = s_red[0], b = s_red[1], d = s_red[2], e = s_red[3];
            s = make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
        }
    }
    if (rl == 0 && c < n) {
        const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c + j;
            if (col >= n) break;
            float* dst = col < n0 ? out0 + col : (out1 ? out1 + (col - n0) : nullptr);
            if (dst) *dst = accumulate ? *dst + v[j] : v[j];
        }
    }
}

// Many partial rows (>= 64) of whole quads: one workgroup per 32-column slice, 8 column lanes x 32 row lanes -- every row access of a wavefront is
// full 128-byte lines (the row-per-lane walk above touches one 16-byte piece per line: 8x the bytes; the ~1000-row reductions behind the
// LayerNorm / mask-pass / row-dot backward kernels took 21 us each, 41 of them per step on the main stream).  Row lanes keep four loads in
// flight; the 32 row-lane sums are folded in a fixed order through LDS: deterministic.
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float* __restrict__ part, int nblk, long stride, int n0,
                                                                   float* __restrict__ out0, int n1, float* __restrict__ out1, int accumulate) {
    __shared__ float4 s_red[32][8];
    const int n = n0 + n1;
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 32 + cl * 4;
    part += (long)blockIdx.y * nblk * stride;
    out0 += (long)blockIdx.y * n0;
    if (out1) out1 += (long)blockIdx.y * n1;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < n) {
        int b = rl;
        for (; b + 96 < nblk; b += 128) {
            const float4 v0 = *reinterpret_cast<const float4*>(part + (long)b * stride + c);
            const float4 v1 = *reinterpret_cast<const float4*>(part + (long)(b + 32) * stride + c);
            const float4 v2 = *reinterpret_cast<const float4*>(part + (long)(b + 64) * stride + c);
            const float4 v3 = *reinterpret_cast<const float4*>(part + (long)(b + 96) * stride + c);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; b < nblk; b += 32) {
            const float4 v = *reinterpret_cast<const float4*>(part + (long)b * stride + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    s_red[rl][cl] = s;
    __syncthreads();
    if (threadIdx.x < 8 && c < n) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 32; ++r) { const float4 v = s_red[r][cl]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c + j;
            float* dst = col < n0 ? out0 + col : (out1 ? out1 + (col - n0) : nullptr);
            if (dst) *dst = accumulate ? *dst + v[j] : v[j];
        }
    }
}

int launch_reduce(const float* part, int batches, int nblk, long stride, int n0, float* out0, int n1, float* out1, int accumulate, hipStream_t stream) {
    const int n = n0 + n1, nq = (n + 3) / 4;
    const int vec = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(part) & 15) == 0) && (((long)nblk * stride) % 4 == 0);
    static const bool wide = [] { const char* e = getenv("MDVIT_REDUCE_WIDE"); return !(e && e[0] == '0'); }();
    if (wide && vec && nblk >= 64 && n % 4 == 0)
        hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3((n + 31) / 32, batches), dim3(256), 0, stream, part, nblk, stride, n0, out0, n1, out1, accumulate);
    else if (nblk >= 256)
        hipLaunchKernelGGL((reduce_partials_kernel<256>), dim3(nq, batches), dim3(256), 0, stream, part, nblk, stride, n0, out0, n1, out1, accumulate, vec);
    else if (nblk >= 16)
        hipLaunchKernelGGL((reduce_partials_kernel<32>), dim3((nq + 7) / 8, batches), dim3(256), 0, stream, part, nblk, stride, n0, out0, n1, out1, accumulate, vec);
    else
        hipLaunchKernelGGL((reduce_partials_kernel<4>), dim3((nq + 63) / 64, batches), dim3(256), 0, stream, part, nblk, stride, n0, out0, n1, out1, accumulate, vec);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
}  // namespace

int mdvit_reduce_partials(const float* part, int nblk, long stride, int n0, float* out0, int n1, float* out1, int accumulate, hipStream_t stream) {
    if (n0 + n1 <= 0 || nblk <= 0) return MDVIT_OK;
    return launch_reduce(part, 1, nblk, stride, n0, out0, n1, out1, accumulate, stream);
}

int mdvit_reduce_partials_batched(const float* part, int batches, int nblk, int n, float* out, hipStream_t stream) {
    if (n <= 0 || nblk <= 0 || batches <= 0) return MDVIT_OK;
    return launch_reduce(part, batches, nblk, (long)n, n, out, 0, nullptr, 0, stream);
}

int mdvit_reduce_partials_batched2(const float* part, int batches, int nblk, int n0, float* out0, int n1, float* out1, hipStream_t stream) {
    if (n0 + n1 <= 0 || nblk <= 0 || batches <= 0) return MDVIT_OK;
    return launch_reduce(part, batches, nblk, (long)(n0 + n1), n0, out0, n1, out1, 0, stream);
}

int mdvit_reduce_partials_batched2_acc(const float* part, int batches, int nblk, int n0, float* out0, int n1, float* out1, int accumulate, hipStream_t stream) {
    if (n0 + n1 <= 0 || nblk <= 0 || batches <= 0) return MDVIT_OK;
    return launch_reduce(part, batches, nblk, (long)(n0 + n1), n0, out0, n1, out1, accumulate, stream);
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

// ---- peer-head weight composition, bias part (decode.py: bc = W_fuse,q b_q for every head and scale; its gradients) -------------------------------------
// n <= 16 items in ONE launch each way (they were 16 rowdot launches forward, 16 + 16 second stages backward).  Fixed summation orders: deterministic.
namespace {
constexpr int CB_MAX = 16;
struct ComposeBiasArgs {
    const float* W[CB_MAX]; const float* b[CB_MAX]; const float* dout[CB_MAX]; float* out[CB_MAX]; float* dW[CB_MAX]; float* db[CB_MAX];
    long ldw, lddw; int rows, cols, db_accumulate;
};
// out[r] = sum_c W[r][c] b[c]: a wavefront per row, lanes stride the columns, xor-tree fold
__global__ __launch_bounds__(256) void compose_bias_fwd_kernel(ComposeBiasArgs p) {
    const int it = blockIdx.y, lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    const float* w = p.W[it] + (long)r * p.ldw;
    const float* b = p.b[it];
    float s = 0.f;
    for (int c = lane; c < p.cols; c += 64) s = fmaf(w[c], b[c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) p.out[it][r] = s;
}
// db[c] (+)= sum_r W[r][c] dout[r] (thread per column, rows in order);   dW[r][c] += dout[r] b[c] (grid.x beyond the column blocks: one row block each)
__global__ __launch_bounds__(256) void compose_bias_bwd_kernel(ComposeBiasArgs p, int col_blocks) {
    const int it = blockIdx.y;
    if ((int)blockIdx.x < col_blocks) {
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c >= p.cols || p.db[it] == nullptr) return;
        const float* w = p.W[it] + c;
        const float* d = p.dout[it];
        float s = 0.f;
        for (int r = 0; r < p.rows; ++r) s = fmaf(w[(long)r * p.ldw], d[r], s);
        p.db[it][c] = p.db_accumulate ? p.db[it][c] + s : s;
        return;
    }
    if (p.dW[it] == nullptr) return;
    const int r0 = ((int)blockIdx.x - col_blocks) * 8;
    for (int r = r0; r < min(r0 + 8, p.rows); ++r) {
        const float dr = p.dout[it][r];
        float* o = p.dW[it] + (long)r * p.lddw;
        for (int c = threadIdx.x; c < p.cols; c += 256) o[c] = fmaf(dr, p.b[it][c], o[c]);
    }
}
}  // namespace

extern "C" int mdvit_compose_bias(int32_t n, const void* const* W, int64_t ldw, const void* const* b, void* const* out, int32_t rows, int32_t cols, void* stream) {
    MDVIT_CHECK_ARG(n >= 1 && n <= CB_MAX && rows > 0 && cols > 0 && W && b && out, MDVIT_E_SHAPE, "compose_bias: 1 <= n <= %d items", CB_MAX);
    ComposeBiasArgs a; memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) { a.W[i] = (const float*)W[i]; a.b[i] = (const float*)b[i]; a.out[i] = (float*)out[i]; }
    a.ldw = ldw; a.rows = rows; a.cols = cols;
    hipLaunchKernelGGL(compose_bias_fwd_kernel, dim3(cdiv(rows, 4), n), dim3(256), 0, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_compose_bias_bwd(int32_t n, const void* const* W, int64_t ldw, const void* const* b, const void* const* dout, void* const* dW, int64_t lddw,
                                      void* const* db, int32_t db_accumulate, int32_t rows, int32_t cols, void* stream) {
    MDVIT_CHECK_ARG(n >= 1 && n <= CB_MAX && rows > 0 && cols > 0 && W && b && dout, MDVIT_E_SHAPE, "compose_bias_bwd: 1 <= n <= %d items", CB_MAX);
    ComposeBiasArgs a; memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) {
        a.W[i] = (const float*)W[i]; a.b[i] = (const float*)b[i]; a.dout[i] = (const float*)dout[i];
        a.dW[i] = dW ? (float*)dW[i] : nullptr; a.db[i] = db ? (float*)db[i] : nullptr;
    }
    a.ldw = ldw; a.lddw = lddw; a.rows = rows; a.cols = cols; a.db_accumulate = db_accumulate;
    const int cb = cdiv(cols, 256);
    hipLaunchKernelGGL(compose_bias_bwd_kernel, dim3(cb + cdiv(rows, 8), n), dim3(256), 0, (hipStream_t)stream, a, cb);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
