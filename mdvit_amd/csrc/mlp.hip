// Fused MLP forward for the C = 64 stages (Mlp.forward mpvit.py:71-78 inside SerialBlock, mdvit.py:357-360):
//     y = res + rowscale * drop2( gelu_drop1(x W1^T + b1) W2^T + b2 ),      h = gelu_drop1(.) also written (the backward's operand)
// One workgroup owns 64 tokens and walks the hidden axis in chunks of 64: u-chunk = x W1c^T on the matrix cores, bias + erf-GELU +
// dropout in registers, the h-chunk goes to HBM once (for the backward) and -- as bf16 hi/lo planes through LDS -- straight into the
// second product y += h-chunk W2c^T.  Against the two-GEMM form this drops the re-read of h [tokens, hidden] by fc2 (42 % of the
// forward MLP traffic at hidden = 8 C) and one launch; the pre-activation u never exists in HBM (the backward recomputes it,
// EPI_DGELU_RC in gemm.hip).  Arithmetic is the bf16x3 GEMM's, element for element: operands split hi + lo, hi*lo + lo*hi + hi*hi
// on v_mfma_f32_32x32x16_bf16, k ascending in slabs of 32 -- so y and h are bit-identical to the two-GEMM path.
#include "common.h"

typedef float mlp_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 mlp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mlp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float mlp_f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int LDKB = 80;           // bytes per [row][32 x bf16] LDS row (64 + 16 pad: conflict-free ds_read_b128 fragments)

__device__ __forceinline__ void mlp_split(const float4 x, uint2& hi, uint2& lo) {
    mlp_f32x2 a = {x.x, x.y}, b = {x.z, x.w};
    const mlp_bf16x2 ha = __builtin_convertvector(a, mlp_bf16x2), hb = __builtin_convertvector(b, mlp_bf16x2);
    const uint32_t hau = __builtin_bit_cast(uint32_t, ha), hbu = __builtin_bit_cast(uint32_t, hb);
    mlp_f32x2 la = {x.x - __uint_as_float(hau << 16), x.y - __uint_as_float(hau & 0xffff0000u)};
    mlp_f32x2 lb = {x.z - __uint_as_float(hbu << 16), x.w - __uint_as_float(hbu & 0xffff0000u)};
    const mlp_bf16x2 lab = __builtin_convertvector(la, mlp_bf16x2), lbb = __builtin_convertvector(lb, mlp_bf16x2);
    hi = make_uint2(hau, hbu);
    lo = make_uint2(__builtin_bit_cast(uint32_t, lab), __builtin_bit_cast(uint32_t, lbb));
}

struct MlpArgs {
    const float* x; const float* W1; const float* b1; const float* W2; const float* b2; const float* res; const float* rowscale;
    float* h; float* y;
    int M, Hd, rows_per_scale;
    int drop; uint32_t k1a, k1b, k2a, k2b, thresh; float inv_keep;
    const uint32_t* seed;
    int abl;                       // diagnosis only (tools/mlp_check.py): 1 no h store, 2 no weight reloads, 4 no GELU, 8 no second product
};

// LDS operand: `slabs` K-slabs of 32, each slab = hi plane [rows][LDKB] then lo plane [rows][LDKB]
__device__ __forceinline__ char* plane(char* base, int rows, int slab, int lo_plane) { return base + ((slab * 2 + lo_plane) * rows) * LDKB; }

// BM tokens per workgroup (BM / 16 threads: 64 -> 4 waves, 2 workgroups per CU; 128 -> 8 waves, 1 workgroup per CU).  Every workgroup
// streams ALL of W1 and W2 (256 KB at hidden = 512) from L2 through its LDS: the wider tile halves that traffic per token.
template <int C, int BM>
__global__ __launch_bounds__(BM * 4) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_fwd_kernel(MlpArgs p) {
    constexpr int HC = 64, NT = BM * 4;             // hidden chunk, threads
    constexpr int S1 = C / 32, S2 = HC / 32;        // K slabs of the two products
    static_assert(C == 64, "tile mapping below is written for C = 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                                 // [S1][2][BM][LDKB]
    char* sW1 = sX + S1 * 2 * BM * LDKB;             // [S1][2][HC][LDKB]
    char* sH = sW1 + S1 * 2 * HC * LDKB;             // [S2][2][BM][LDKB]
    char* sW2 = sH + S2 * 2 * BM * LDKB;             // [S2][2][C][LDKB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1, k2a = p.k2a ^ s0, k2b = p.k2b + s1;

    // stage a [ROWS][64 k] fp32 tile (row-major, leading dimension ld) as bf16 hi/lo planes: ROWS * 16 / NT float4 per thread
    constexpr int XV = BM * 16 / NT, WV = 64 * 16 / NT;
    float4 rx[XV], r1[WV], r2[WV];
    auto load_tile = [&](auto& r, const float* src, long ld, int row0, int nrows) {
        constexpr int NV = sizeof(r) / sizeof(float4);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            // UNCONDITIONAL (rows past the end re-read the last row: they only feed outputs that are never stored): a load under a
            // branch makes hipcc drain the whole vector-memory queue at the join, which defeats the weight prefetch below
            const int idx = tid + NT * v, row = idx >> 4, c4 = idx & 15;
            r[v] = *reinterpret_cast<const float4*>(src + (long)min(row0 + row, nrows - 1) * ld + c4 * 4);
        }
    };
    auto store_tile = [&](const auto& r, char* dst, int rows) {
        constexpr int NV = sizeof(r) / sizeof(float4);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + NT * v, row = idx >> 4, k = (idx & 15) * 4;
            uint2 hi, lo;
            mlp_split(r[v], hi, lo);
            *reinterpret_cast<uint2*>(plane(dst, rows, k >> 5, 0) + row * LDKB + (k & 31) * 2) = hi;
            *reinterpret_cast<uint2*>(plane(dst, rows, k >> 5, 1) + row * LDKB + (k & 31) * 2) = lo;
        }
    };
    // one 64-deep product of a wave's 32x32 block: A rows = tokens (wm0 + l31), B rows = output columns (wn0 + l31)
    auto product = [&](const char* A, int arows, const char* B, mlp_f32x16& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int koff = (2 * ks + lhi) * 16;
                const mlp_bf16x8 ah = __builtin_bit_cast(mlp_bf16x8, *reinterpret_cast<const uint4*>(plane(const_cast<char*>(A), arows, s, 0) + (wm0 + l31) * LDKB + koff));
                const mlp_bf16x8 al = __builtin_bit_cast(mlp_bf16x8, *reinterpret_cast<const uint4*>(plane(const_cast<char*>(A), arows, s, 1) + (wm0 + l31) * LDKB + koff));
                const mlp_bf16x8 bh = __builtin_bit_cast(mlp_bf16x8, *reinterpret_cast<const uint4*>(plane(const_cast<char*>(B), 64, s, 0) + (wn0 + l31) * LDKB + koff));
                const mlp_bf16x8 bl = __builtin_bit_cast(mlp_bf16x8, *reinterpret_cast<const uint4*>(plane(const_cast<char*>(B), 64, s, 1) + (wn0 + l31) * LDKB + koff));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc, 0, 0, 0);
            }
    };

    // x tile (resident for the whole walk) and the first weight chunks
    load_tile(rx, p.x, C, m0, p.M);
    store_tile(rx, sX, BM);
    load_tile(r1, p.W1, C, 0, p.Hd);                       // W1 rows [0, 64), all C columns
    load_tile(r2, p.W2, p.Hd, 0, C);                       // W2 rows = all C outputs, columns [0, 64)
    mlp_f32x16 yacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[r] = 0.f;
    const int row = m0 + wm0 + l31;                        // the token row this lane's accumulator quads belong to

    for (int hc0 = 0; hc0 < p.Hd; hc0 += HC) {
        store_tile(r1, sW1, 64);
        store_tile(r2, sW2, 64);
        __syncthreads();
        // this chunk's bias quads FIRST, then the next chunk's weights (the last iteration re-reads its own chunk): the epilogue
        // below then waits for the OLDER loads only (counted vmcnt) and the weight prefetch stays in flight behind the arithmetic.
        // (It used to read b1 in the epilogue: in-order vmcnt made every chunk wait for the prefetch it had just issued.)
        float4 b1q[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b1q[q] = *reinterpret_cast<const float4*>(p.b1 + hc0 + wn0 + 8 * q + 4 * lhi);
        __builtin_amdgcn_sched_barrier(0);          // (keeps the bias loads OLDER than the prefetch in the in-order vmcnt queue)
        {
            const int hn = min(hc0 + HC, p.Hd - HC);
            load_tile(r1, p.W1 + (long)hn * C, C, 0, HC);
            load_tile(r2, p.W2 + hn, p.Hd, 0, C);
        }
        __builtin_amdgcn_sched_barrier(0);
        mlp_f32x16 uacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) uacc[r] = 0.f;
        product(sX, BM, sW1, uacc);
        // bias + GELU + dropout; h to HBM (float4 per quad) and to the LDS operand planes of the second product
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = wn0 + 8 * q + 4 * lhi, hd = hc0 + col;
            const float4 b4 = b1q[q];
            float4 hv = (p.abl & 4) ? make_float4(uacc[4 * q + 0] + b4.x, uacc[4 * q + 1] + b4.y, uacc[4 * q + 2] + b4.z, uacc[4 * q + 3] + b4.w)
                                    : make_float4(gelu_f(uacc[4 * q + 0] + b4.x), gelu_f(uacc[4 * q + 1] + b4.y),
                                                  gelu_f(uacc[4 * q + 2] + b4.z), gelu_f(uacc[4 * q + 3] + b4.w));
            if (p.drop) {
                const float4 ds = mdvit_drop_scale4(k1a, k1b, (uint32_t)((long)row * p.Hd + hd), p.thresh, p.inv_keep);
                hv.x *= ds.x; hv.y *= ds.y; hv.z *= ds.z; hv.w *= ds.w;
            }
            if (row < p.M && !(p.abl & 1)) *reinterpret_cast<float4*>(p.h + (long)row * p.Hd + hd) = hv;
            uint2 hi, lo;
            mlp_split(hv, hi, lo);
            *reinterpret_cast<uint2*>(plane(sH, BM, col >> 5, 0) + (wm0 + l31) * LDKB + (col & 31) * 2) = hi;
            *reinterpret_cast<uint2*>(plane(sH, BM, col >> 5, 1) + (wm0 + l31) * LDKB + (col & 31) * 2) = lo;
        }
        __syncthreads();
        if (!(p.abl & 8)) product(sH, BM, sW2, yacc);
        __syncthreads();                                   // sW1 / sW2 / sH are rewritten by the next chunk
    }

    {
        // all epilogue operands requested at once (one exposed latency instead of eight): bias and residual quads, the row's DropPath scale
        const int rowc = min(row, p.M - 1);
        float4 b2q[4], rq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = wn0 + 8 * q + 4 * lhi;
            b2q[q] = *reinterpret_cast<const float4*>(p.b2 + col);
            rq[q] = *reinterpret_cast<const float4*>(p.res + (long)rowc * C + col);
        }
        const float rsc = p.rowscale ? p.rowscale[rowc / p.rows_per_scale] : 1.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = wn0 + 8 * q + 4 * lhi;
            const float4 b4 = b2q[q];
            float4 v = make_float4(yacc[4 * q + 0] + b4.x, yacc[4 * q + 1] + b4.y, yacc[4 * q + 2] + b4.z, yacc[4 * q + 3] + b4.w);
            if (p.drop) {
                const float4 ds = mdvit_drop_scale4(k2a, k2b, (uint32_t)((long)row * C + col), p.thresh, p.inv_keep);
                v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
            }
            v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
            const float4 r4 = rq[q];
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            if (row < p.M) *reinterpret_cast<float4*>(p.y + (long)row * C + col) = v;
        }
    }
}

// ---- backward data path of the same MLP:  dx = ((gm W2) * gelu'(u) * dropmask1) W1,  u = x W1^T + b1 recomputed per chunk ----
// gm = the masked upstream gradient (mdvit_colsum_f32).  A workgroup owns 64 tokens; the MFMA A-fragments of its x and gm tiles live
// in registers for the whole walk (they are the A operands of two K = C products per hidden chunk); per chunk: u = x W1c^T and
// d = gm W2c on the matrix cores, d *= gelu'(u + b1) * mask in registers, d to HBM only if the caller wants it (the weight
// gradients of the full sweep do; the data-gradient-only sweep does not -- then [tokens, hidden] never touches HBM), d through LDS
// into dx += d W1c.  Same products, slab order and keys as the recomputing GEMM + fc1 data-gradient GEMM it replaces.
struct MlpBwdArgs {
    const float* gm; const float* x; const float* W1; const float* b1; const float* W2t; const float* W1t;
    float* du; float* dx;
    int M, Hd;
    int drop; uint32_t k1a, k1b, thresh; float inv_keep;
    const uint32_t* seed;
};

template <int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_bwd_dgrad_kernel(MlpBwdArgs p) {
    constexpr int BM = 64, HC = 64;
    static_assert(C == 64, "tile mapping below is written for C = 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE = 2 * 2 * 64 * LDKB;          // one [64 rows][64 k] operand: 2 slabs x (hi, lo) planes
    char* sW1 = smem;                                // W1 rows [hc0, hc0+64) x k = c            (B of u = x W1c^T)
    char* sW2t = sW1 + TILE;                         // W2^T rows [hc0, hc0+64) x k = c          (B of d = gm W2c)
    char* sW1t = sW2t + TILE;                        // W1^T rows c x k = [hc0, hc0+64)          (B of dx += d W1c)
    char* sD = sW1t + TILE;                          // d chunk: rows = tokens x k = hidden chunk (A of the third product)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1;

    float4 r1[4], r2[4], r3[4];
    auto load_tile = [&](float4 (&r)[4], const float* src, long ld, int row0, int nrows) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int idx = tid + 256 * v, row = idx >> 4, c4 = idx & 15;          // unconditional, see mlp_fwd_kernel
            r[v] = *reinterpret_cast<const float4*>(src + (long)min(row0 + row, nrows - 1) * ld + c4 * 4);
        }
    };
    auto store_tile = [&](const float4 (&r)[4], char* dst) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int idx = tid + 256 * v, row = idx >> 4, k = (idx & 15) * 4;
            uint2 hi, lo;
            mlp_split(r[v], hi, lo);
            *reinterpret_cast<uint2*>(plane(dst, 64, k >> 5, 0) + row * LDKB + (k & 31) * 2) = hi;
            *reinterpret_cast<uint2*>(plane(dst, 64, k >> 5, 1) + row * LDKB + (k & 31) * 2) = lo;
        }
    };
    auto frag = [&](char* base, int s, int lo_plane, int rowi, int ks) {
        return __builtin_bit_cast(mlp_bf16x8, *reinterpret_cast<const uint4*>(plane(base, 64, s, lo_plane) + rowi * LDKB + (2 * ks + lhi) * 16));
    };

    // x and gm tiles -> LDS (through the W1 / W2t regions) -> this wave's A-fragments, kept in registers
    load_tile(r1, p.x, C, m0, p.M);
    load_tile(r2, p.gm, C, m0, p.M);
    store_tile(r1, sW1);
    store_tile(r2, sW2t);
    __syncthreads();
    mlp_bf16x8 xh[4], xl[4], gh[4], gl[4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            xh[2 * s + ks] = frag(sW1, s, 0, wm0 + l31, ks); xl[2 * s + ks] = frag(sW1, s, 1, wm0 + l31, ks);
            gh[2 * s + ks] = frag(sW2t, s, 0, wm0 + l31, ks); gl[2 * s + ks] = frag(sW2t, s, 1, wm0 + l31, ks);
        }
    __syncthreads();
    load_tile(r1, p.W1, C, 0, p.Hd);
    load_tile(r2, p.W2t, C, 0, p.Hd);
    load_tile(r3, p.W1t, p.Hd, 0, C);
    mlp_f32x16 dxacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dxacc[r] = 0.f;
    const int row = m0 + wm0 + l31;

    for (int hc0 = 0; hc0 < p.Hd; hc0 += HC) {
        store_tile(r1, sW1);
        store_tile(r2, sW2t);
        store_tile(r3, sW1t);
        __syncthreads();
        float4 b1q[4];                      // bias quads before the prefetch: see mlp_fwd_kernel
#pragma unroll
        for (int q = 0; q < 4; ++q) b1q[q] = *reinterpret_cast<const float4*>(p.b1 + hc0 + wn0 + 8 * q + 4 * lhi);
        __builtin_amdgcn_sched_barrier(0);          // (keeps the bias loads OLDER than the prefetch in the in-order vmcnt queue)
        {
            const int hn = min(hc0 + HC, p.Hd - HC);
            load_tile(r1, p.W1 + (long)hn * C, C, 0, HC);
            load_tile(r2, p.W2t + (long)hn * C, C, 0, HC);
            load_tile(r3, p.W1t + hn, p.Hd, 0, C);
        }
        __builtin_amdgcn_sched_barrier(0);
        mlp_f32x16 uacc, dacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) { uacc[r] = 0.f; dacc[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const mlp_bf16x8 bh = frag(sW1, s, 0, wn0 + l31, ks), bl = frag(sW1, s, 1, wn0 + l31, ks);
                uacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, xh[2 * s + ks], uacc, 0, 0, 0);
                uacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, xl[2 * s + ks], uacc, 0, 0, 0);
                uacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, xh[2 * s + ks], uacc, 0, 0, 0);
                const mlp_bf16x8 ch = frag(sW2t, s, 0, wn0 + l31, ks), cl = frag(sW2t, s, 1, wn0 + l31, ks);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl, gh[2 * s + ks], dacc, 0, 0, 0);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, gl[2 * s + ks], dacc, 0, 0, 0);
                dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch, gh[2 * s + ks], dacc, 0, 0, 0);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = wn0 + 8 * q + 4 * lhi, hd = hc0 + col;
            const float4 b4 = b1q[q];
            float4 dv = make_float4(dacc[4 * q + 0] * gelu_grad_f(uacc[4 * q + 0] + b4.x), dacc[4 * q + 1] * gelu_grad_f(uacc[4 * q + 1] + b4.y),
                                    dacc[4 * q + 2] * gelu_grad_f(uacc[4 * q + 2] + b4.z), dacc[4 * q + 3] * gelu_grad_f(uacc[4 * q + 3] + b4.w));
            if (p.drop) {
                const float4 ds = mdvit_drop_scale4(k1a, k1b, (uint32_t)((long)row * p.Hd + hd), p.thresh, p.inv_keep);
                dv.x *= ds.x; dv.y *= ds.y; dv.z *= ds.z; dv.w *= ds.w;
            }
            if (p.du && row < p.M) *reinterpret_cast<float4*>(p.du + (long)row * p.Hd + hd) = dv;
            uint2 hi, lo;
            mlp_split(dv, hi, lo);
            *reinterpret_cast<uint2*>(plane(sD, 64, col >> 5, 0) + (wm0 + l31) * LDKB + (col & 31) * 2) = hi;
            *reinterpret_cast<uint2*>(plane(sD, 64, col >> 5, 1) + (wm0 + l31) * LDKB + (col & 31) * 2) = lo;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const mlp_bf16x8 ah = frag(sD, s, 0, wm0 + l31, ks), al = frag(sD, s, 1, wm0 + l31, ks);
                const mlp_bf16x8 bh = frag(sW1t, s, 0, wn0 + l31, ks), bl = frag(sW1t, s, 1, wn0 + l31, ks);
                dxacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, dxacc, 0, 0, 0);
                dxacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, dxacc, 0, 0, 0);
                dxacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, dxacc, 0, 0, 0);
            }
        __syncthreads();
    }
    if (row < p.M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = wn0 + 8 * q + 4 * lhi;
            *reinterpret_cast<float4*>(p.dx + (long)row * C + col) = make_float4(dxacc[4 * q + 0], dxacc[4 * q + 1], dxacc[4 * q + 2], dxacc[4 * q + 3]);
        }
    }
}

}  // namespace

int g_mlp_wide_min_tokens = 32768;       // tokens from which the 128-token tile is used (tuning hook: mdvit_mlp_config)
int g_mlp_abl = 0;

extern "C" int mdvit_mlp_config(int32_t wide_min_tokens, int32_t ablate) {
    g_mlp_wide_min_tokens = wide_min_tokens > 0 ? wide_min_tokens : 0x7fffffff;
    g_mlp_abl = ablate;
    return MDVIT_OK;
}

extern "C" int mdvit_mlp_fwd_f32(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* res,
                                 const float* rowscale, int32_t rows_per_scale, float* h, float* y, int32_t M, int32_t C, int32_t Hd,
                                 float drop_p, uint32_t key1_0, uint32_t key1_1, uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed,
                                 void* stream) {
    MDVIT_CHECK_ARG(C == 64, MDVIT_E_SHAPE, "mlp_fwd: built for C = 64 (got %d)", C);
    MDVIT_CHECK_ARG(M > 0 && Hd > 0 && Hd % 64 == 0, MDVIT_E_SHAPE, "mlp_fwd: need M > 0, hidden %% 64 == 0 (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(x && W1 && b1 && W2 && b2 && res && h && y, MDVIT_E_SHAPE, "mlp_fwd: null operand");
    MDVIT_CHECK_ARG(aligned16(x) && aligned16(W1) && aligned16(b1) && aligned16(W2) && aligned16(b2) && aligned16(res) && aligned16(h) && aligned16(y),
                    MDVIT_E_ALIGN, "mlp_fwd: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp_fwd: dropout index space exceeds 2^32");
    MlpArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.res = res; a.rowscale = rowscale; a.h = h; a.y = y;
    a.M = M; a.Hd = Hd; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    a.drop = drop_p > 0.f; a.k1a = key1_0; a.k1b = key1_1; a.k2a = key2_0; a.k2b = key2_1;
    a.thresh = mdvit_drop_thresh(drop_p); a.inv_keep = 1.f / (1.f - drop_p);
    a.seed = drop_seed;
    a.abl = g_mlp_abl;
    constexpr size_t smem64 = (size_t)(2 + 2 + 2 + 2) * 2 * 64 * LDKB;         // X, W1c, Hc, W2c at 64 tokens: 80 KB
    constexpr size_t smem128 = (size_t)(4 + 2 + 4 + 2) * 2 * 64 * LDKB;        // 128 tokens: X and Hc double: 120 KB
    {   // the attribute is per DEVICE: one flag per device ordinal (a process may drive several GPUs)
        static bool attr_set[64] = {false};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_set[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_kernel<64, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem64);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_kernel<64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
            if (e != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "mlp_fwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            attr_set[dev] = true;
        }
    }
    if (M >= g_mlp_wide_min_tokens) hipLaunchKernelGGL((mlp_fwd_kernel<64, 128>), dim3(cdiv(M, 128)), dim3(512), smem128, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_fwd_kernel<64, 64>), dim3(cdiv(M, 64)), dim3(256), smem64, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_mlp_bwd_dgrad_f32(const float* gm, const float* x, const float* W1, const float* b1, const float* W2t, const float* W1t,
                                       float* du, float* dx, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1,
                                       const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(C == 64, MDVIT_E_SHAPE, "mlp_bwd_dgrad: built for C = 64 (got %d)", C);
    MDVIT_CHECK_ARG(M > 0 && Hd > 0 && Hd % 64 == 0, MDVIT_E_SHAPE, "mlp_bwd_dgrad: need M > 0, hidden %% 64 == 0 (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(gm && x && W1 && b1 && W2t && W1t && dx, MDVIT_E_SHAPE, "mlp_bwd_dgrad: null operand");
    MDVIT_CHECK_ARG(aligned16(gm) && aligned16(x) && aligned16(W1) && aligned16(b1) && aligned16(W2t) && aligned16(W1t) && aligned16(dx) &&
                    (!du || aligned16(du)), MDVIT_E_ALIGN, "mlp_bwd_dgrad: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp_bwd_dgrad: dropout index space exceeds 2^32");
    MlpBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.gm = gm; a.x = x; a.W1 = W1; a.b1 = b1; a.W2t = W2t; a.W1t = W1t; a.du = du; a.dx = dx; a.M = M; a.Hd = Hd;
    a.drop = drop_p > 0.f; a.k1a = key1_0; a.k1b = key1_1;
    a.thresh = mdvit_drop_thresh(drop_p); a.inv_keep = 1.f / (1.f - drop_p);
    a.seed = drop_seed;
    constexpr size_t smem = (size_t)4 * 2 * 2 * 64 * LDKB;                    // W1c, W2tc, W1tc, d chunk: 80 KB
    {
        static bool attr_set[64] = {false};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!attr_set[dev]) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_dgrad_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "mlp_bwd_dgrad: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL((mlp_bwd_dgrad_kernel<64>), dim3(cdiv(M, 64)), dim3(256), smem, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
