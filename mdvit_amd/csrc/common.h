// Shared device/host helpers for libmdvit_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mdvit_hip.h"

#define MDVIT_WAVE 64

// ---- thread-local last-error string --------------------------------------------------------
extern thread_local char g_mdvit_err[512];
int mdvit_set_error(int code, const char* fmt, ...);

#define MDVIT_CHECK_ARG(cond, code, ...)                      \
    do {                                                      \
        if (!(cond)) return mdvit_set_error(code, __VA_ARGS__); \
    } while (0)

// Kernel-execution timing for bench.py's roofline: mdvit_timing_arm(start, stop) makes the NEXT GEMM main-kernel launch record its
// own begin / end timestamps into the two events (hipExtLaunchKernelGGL) -- the dispatch's duration as rocprofv3 reports it, without
// the time the launch waits for CUs held by the other streams' kernels, which a record-before / record-after event pair includes.
extern hipEvent_t g_mdvit_t0, g_mdvit_t1;       // abi.hip
#define MDVIT_TIMED_LAUNCH(kernel, grid, block, shmem, s, ...)                                                       \
    do {                                                                                                              \
        if (g_mdvit_t0) {                                                                                             \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, s, g_mdvit_t0, g_mdvit_t1, 0, __VA_ARGS__);             \
            g_mdvit_t0 = g_mdvit_t1 = nullptr;                                                                        \
        } else {                                                                                                      \
            hipLaunchKernelGGL(kernel, grid, block, shmem, s, __VA_ARGS__);                                           \
        }                                                                                                             \
    } while (0)

#define MDVIT_LAUNCH_CHECK()                                                         \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess)                                                       \
            return mdvit_set_error(MDVIT_E_HIP, "%s:%d launch failed: %s", __FILE__, \
                                   __LINE__, hipGetErrorString(e__));                \
    } while (0)

// Zero-fill as an ordinary kernel on the launch stream (strictly stream-ordered with the kernels around it).
int mdvit_zero_async(void* ptr, size_t bytes, hipStream_t stream);
// Zero several buffers; runs of buffers that are adjacent in memory are cleared by one launch (callers allocate
// the small gradient outputs of one op as slices of a single buffer).
struct MdvitZeroItem { void* p; size_t bytes; };
int mdvit_zero_many(const MdvitZeroItem* items, int n, hipStream_t stream);
#define MDVIT_ZERO(ptr, bytes, strm)                                   \
    do {                                                               \
        int rc__ = mdvit_zero_async((ptr), (bytes), (strm));           \
        if (rc__ != MDVIT_OK) return rc__;                             \
    } while (0)

// out[i] (+)= sum_{b < nblk} part[b * stride + i], i < n, added in a FIXED order (deterministic): the second stage of
// "every workgroup writes one row of partial sums".  Same-address float atomics from ~1000 workgroups serialise in the
// L2 (a reduction's atomic tail cost 30-100 us); a partial row per workgroup plus this ~5 us pass does not.
// Row layout [out0 (n0) | out1 (n1)]; out1 may be NULL (then n1 entries are skipped).
int mdvit_reduce_partials(const float* part, int nblk, long stride, int n0, float* out0, int n1, float* out1, int accumulate, hipStream_t stream);
// `batches` independent reductions: part [batch][nblk][n] -> out [batch][n]
int mdvit_reduce_partials_batched(const float* part, int batches, int nblk, int n, float* out, hipStream_t stream);
// batched with two outputs per batch: out0 [batch][n0], out1 [batch][n1]; partial rows are [n0 | n1] wide
int mdvit_reduce_partials_batched2(const float* part, int batches, int nblk, int n0, float* out0, int n1, float* out1, hipStream_t stream);
int mdvit_reduce_partials_batched2_acc(const float* part, int batches, int nblk, int n0, float* out0, int n1, float* out1, int accumulate, hipStream_t stream);
// first stages alone (norm.hip): the partial rows stay in `ws`, *nblk of them per batch; the caller runs the second stage on a stream of its choice
int mdvit_layernorm_bwd_parts(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* add, float* dx,
                              float* dx_masked, void* ws, size_t ws_bytes, int M, int C, int groups, float drop_p, uint32_t key0, uint32_t key1,
                              const float* rowscale, int rows_per_scale, const uint32_t* seed, hipStream_t stream, int* nblk);
int mdvit_colsum_parts(const float* A, long lda, float* masked, void* ws, size_t ws_bytes, int M, int N, float drop_p, uint32_t key0, uint32_t key1,
                       const float* rowscale, int rows_per_scale, const uint32_t* seed, hipStream_t stream, int* nblk);
// the NEXT LayerNorm backward call of this thread reads its upstream gradient as dy + dy2 (norm.hip; consumed by that call)
void mdvit_layernorm_bwd_next_dy2(const float* dy2);
constexpr int MDVIT_MAX_PARTIAL_ROWS = 2048;        // every partial-row reduction launches at most this many workgroups
#define MDVIT_CHECK_PARTIALS_WS(ws, ws_bytes, nblk, n, what)                                                              \
    MDVIT_CHECK_ARG((ws) != nullptr && (ws_bytes) >= sizeof(float) * (size_t)(nblk) * (size_t)(n), MDVIT_E_WORKSPACE,    \
                    what ": workspace too small: need %zu bytes (mdvit_partials_ws_bytes), got %zu",                      \
                    sizeof(float) * (size_t)(nblk) * (size_t)(n), (size_t)(ws_bytes))

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- counter-based dropout RNG (stateless; the backward pass re-derives the mask) ----------
__device__ __forceinline__ uint32_t mdvit_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// Dropout keep-scales (0 or inv_keep), thresh = round(p * 2^32).  ONE 32-bit hash per aligned group of four elements,
// rotated by a byte per element: each element still sees a uniform 32-bit value (exact drop probability), the four
// decisions hang on disjoint top bytes, and the two quarter-rate v_mul_lo_u32 rounds are paid once per float4 instead
// of once per element (they dominated the VALU time of the K=64/128 MLP GEMM epilogues).
// (One round of the lowbias32 mixer over (quad index ^ k0) + k1.  Round 2 ran it twice; against the two-round form the keep rates, the
//  correlations inside a quad, between neighbouring quads and between related keys are indistinguishable over 4 M indices x 6 key pairs x
//  p in {0.1, 0.25, 0.5} (max |corr| 0.003 either way), and the second round was two more quarter-rate v_mul_lo_u32 per four elements in
//  every dropout epilogue -- 10 % of the VALU time of the fused MLP kernels, which are VALU-bound.)
__device__ __forceinline__ uint32_t mdvit_drop_bits(uint32_t k0, uint32_t k1, uint32_t idx) {
    return mdvit_hash32(((idx >> 2) ^ k0) + k1);
}
// the same word from the group's index idx >> 2 (kernels that step a per-lane group index instead of re-deriving it from (row, column) per group)
__device__ __forceinline__ uint32_t mdvit_drop_bits_q(uint32_t k0, uint32_t k1, uint32_t quad) { return mdvit_hash32((quad ^ k0) + k1); }
__device__ __forceinline__ float mdvit_drop_scale(uint32_t k0, uint32_t k1, uint32_t idx, uint32_t thresh, float inv_keep) {
    const uint32_t h = __builtin_rotateright32(mdvit_drop_bits(k0, k1, idx), 8u * (idx & 3u));
    return h >= thresh ? inv_keep : 0.0f;
}
// The threshold every launcher hands to the kernels: p at 16-bit resolution in the TOP half, low half zero (p = 0.1 -> 6554 / 65536 = 0.100006).
// With the low half zero `h >= thresh` is decided by the top 16 bits of h alone, so a kernel may test `(h >> 16) >= (thresh >> 16)` -- one SDWA compare on
// a word of the hash, no rotate for two of the four elements of a group -- and still draw EXACTLY the mask of a kernel that compares all 32 bits.
static inline uint32_t mdvit_drop_thresh(double p) {
    double t = p * 65536.0 + 0.5;
    if (t < 0.0) t = 0.0;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t << 16;
}
// The four keep decisions of one hash word (elements idx4 .. idx4 + 3 of mdvit_drop_scale4), thresh16 = thresh >> 16: element j looks at the top half of
// rotr(h, 8 j), i.e. word 1 of h, word 1 of rotr(h, 8), word 0 of h, word 0 of rotr(h, 8) -- one v_alignbit + four word compares.
__device__ __forceinline__ void mdvit_drop_keep4(uint32_t h, uint32_t thresh16, bool (&keep)[4]) {
    const uint32_t r = __builtin_rotateright32(h, 8);
    keep[0] = (h >> 16) >= thresh16;
    keep[1] = (r >> 16) >= thresh16;
    keep[2] = (h & 0xffffu) >= thresh16;
    keep[3] = (r & 0xffffu) >= thresh16;
}
// idx4 % 4 == 0: scales of elements idx4 .. idx4+3
__device__ __forceinline__ float4 mdvit_drop_scale4(uint32_t k0, uint32_t k1, uint32_t idx4, uint32_t thresh, float inv_keep) {
    const uint32_t h = mdvit_drop_bits(k0, k1, idx4);
    return make_float4(h >= thresh ? inv_keep : 0.0f,
                       __builtin_rotateright32(h, 8) >= thresh ? inv_keep : 0.0f,
                       __builtin_rotateright32(h, 16) >= thresh ? inv_keep : 0.0f,
                       __builtin_rotateright32(h, 24) >= thresh ? inv_keep : 0.0f);
}

// Workgroup ids go round robin over the 8 XCDs (each with its own L2): consecutive LOGICAL ids on ONE XCD, bijectively.  For kernels whose neighbouring
// workgroups re-read the same lines (bilinear taps, row folds): with the raw id every XCD fetches them from HBM once more.
__device__ __forceinline__ unsigned mdvit_xcd_logical_block() {
    const unsigned nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// LayerNorm pieces with EXPLICIT rounding (no fma contraction left to the compiler): two kernels that normalise the same row -- ln_fwd16_kernel and
// the LayerNorm prologue of lin_rc_kernel -- must agree bit for bit whatever surrounds the expression
__device__ __forceinline__ float mdvit_ln_sq4(const float4 v) {
#pragma clang fp contract(off)
    // every fused multiply-add is written out (HIP's __fmul_rn / __fadd_rn are plain operators: the compiler contracts them as it likes)
    const float yy = v.y * v.y, ww = v.w * v.w;
    return __builtin_fmaf(v.x, v.x, yy) + __builtin_fmaf(v.z, v.z, ww);
}
__device__ __forceinline__ float4 mdvit_ln_affine4(const float4 v, float rs, const float4 g, const float4 b) {
#pragma clang fp contract(off)
    const float tx = v.x * rs, ty = v.y * rs, tz = v.z * rs, tw = v.w * rs;
    return make_float4(__builtin_fmaf(tx, g.x, b.x), __builtin_fmaf(ty, g.y, b.y), __builtin_fmaf(tz, g.z, b.z), __builtin_fmaf(tw, g.w, b.w));
}
// (1 / sqrt as ONE v_rsq_f32, 1 ulp: `1.0f / sqrtf(x)` compiles to different instruction sequences in different translation units)
__device__ __forceinline__ float mdvit_ln_rstd(float sumsq, float inv_c, float eps) { return __builtin_amdgcn_rsqf(__builtin_fmaf(sumsq, inv_c, eps)); }
#define ln_sq4 mdvit_ln_sq4
#define ln_affine4 mdvit_ln_affine4

// ---- bf16 "planes": x = hi + lo with hi = RNE bf16(x), lo = RNE bf16(x - hi)  (|x - hi - lo| <= 2^-18 |x|) -------------------
typedef __bf16 mdvit_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mdvit_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mdvit_split_bf16x3(const float4 x, uint2& hi, uint2& lo) {
    mdvit_f32x2 a = {x.x, x.y}, b = {x.z, x.w};
    const mdvit_bf16x2 ha = __builtin_convertvector(a, mdvit_bf16x2), hb = __builtin_convertvector(b, mdvit_bf16x2);
    const uint32_t hau = __builtin_bit_cast(uint32_t, ha), hbu = __builtin_bit_cast(uint32_t, hb);
    mdvit_f32x2 la = {x.x - __uint_as_float(hau << 16), x.y - __uint_as_float(hau & 0xffff0000u)};
    mdvit_f32x2 lb = {x.z - __uint_as_float(hbu << 16), x.w - __uint_as_float(hbu & 0xffff0000u)};
    const mdvit_bf16x2 lab = __builtin_convertvector(la, mdvit_bf16x2), lbb = __builtin_convertvector(lb, mdvit_bf16x2);
    hi = make_uint2(hau, hbu);
    lo = make_uint2(__builtin_bit_cast(uint32_t, lab), __builtin_bit_cast(uint32_t, lbb));
}
__device__ __forceinline__ void mdvit_split1_bf16x3(float x, uint16_t& hi, uint16_t& lo) {
    mdvit_f32x2 a = {x, 0.f};
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, mdvit_bf16x2));
    mdvit_f32x2 l = {x - __uint_as_float(h << 16), 0.f};
    hi = (uint16_t)(h & 0xffffu);
    lo = (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(l, mdvit_bf16x2)) & 0xffffu);
}
// 4 consecutive elements of a plane pair -> fp32 (hi + lo; lo == nullptr: the bf16 speed mode keeps hi only)
__device__ __forceinline__ float4 mdvit_join_planes4(const uint2 hi, const uint2 lo) {
    return make_float4(__uint_as_float(hi.x << 16) + __uint_as_float(lo.x << 16), __uint_as_float(hi.x & 0xffff0000u) + __uint_as_float(lo.x & 0xffff0000u),
                       __uint_as_float(hi.y << 16) + __uint_as_float(lo.y << 16), __uint_as_float(hi.y & 0xffff0000u) + __uint_as_float(lo.y & 0xffff0000u));
}

// ---- wave / block reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-GELU (nn.GELU default) and its derivative.  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32
// round-off class): erf(z) = 1 - (a1 t + .. + a5 t^5) exp(-z^2), t = 1/(1 + p z).  With z = |x|/sqrt(2) the
// exponential exp(-x^2/2) is shared with the Gaussian pdf of the derivative: ONE v_exp + ONE v_rcp per element
// instead of libm erff + expf (~5x fewer VALU instructions in the GEMM epilogues).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x);
    const float e = __expf(-0.5f * x * x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erf_abs = 1.0f - poly * t * e;           // erf(|x|/sqrt2)
    cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
    pdf = 0.39894228040143268f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return fmaf(x, pdf, cdf);
}
__device__ __forceinline__ float hswish_f(float x) { return x * fminf(fmaxf(x + 3.0f, 0.0f), 6.0f) * (1.0f / 6.0f); }
__device__ __forceinline__ float hswish_grad_f(float x) {
    // d/dx [x * clamp(x+3,0,6)/6]; ATen: 0 for x<-3, 1 for x>3, (2x+3)/6 in between
    return x < -3.0f ? 0.0f : (x > 3.0f ? 1.0f : (2.0f * x + 3.0f) * (1.0f / 6.0f));
}
