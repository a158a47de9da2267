// Fused MLP of the HBM-bound stages whose hidden activation NEVER touches HBM (Mlp.forward mpvit.py:71-78 inside SerialBlock_adapt,
// mdvit.py:357-360, and its autograd backward):
//     forward   y  = res + rowscale * drop2( drop1(gelu(x W1^T + b1)) W2^T + b2 )                          x, res in -> y out
//     dgrad     dx = ((gm W2) * gelu'(x W1^T + b1) * mask1) W1                                             x, gm in  -> dx out
//     wgrad     dW1 = du^T x, db1 = colsum(du), dW2 = gm^T h   with h, du RECOMPUTED from x, gm            x, gm in  -> dW1, db1, dW2 out
// Round 2's kernels (mlp.hip) wrote h [tokens, hidden] in the forward and du [tokens, hidden] in the backward for the two weight-gradient
// GEMMs: 4.3 of the 5.3 GB the MLP of one stage-0 block moves at bs=32.  Here nothing of that size exists: the weight-gradient kernel
// recomputes u, h and du tile by tile from the [tokens, C] operands.
//
// Structure (forward / dgrad): a WAVE owns 32 tokens end to end; the waves of a workgroup share only the weight sub-tiles, which arrive
// PRE-SPLIT (the per-step bf16 planes of mdvit_split_planes_many) by global_load_lds through three-slot rings with counted vmcnt and a raw
// s_barrier per 32-wide hidden step -- no conversion work on weights, two steps of prefetch in flight, no ordinary global load inside the
// loop (biases sit in LDS).  Product 1 runs as D[hidden][token], so a lane holds 16 hidden values of ITS token: after bias / GELU /
// dropout / hi-lo split they ARE product 2's operand, in a permuted k order that the weight operand is read in as well (rc_frag_perm;
// round 3 restored the natural order with v_permlane32_swap -- the GEMM's summation order bit for bit, at 8 slow VALU instructions per
// step) -- the hidden chunk never leaves the register file and the arithmetic is the bf16x3 GEMM's products (y equals mdvit_mlp_fwd_f32 /
// the two-GEMM path to a few ulp).  Round 5: on one SIMD an MFMA does not run beside a saturated VALU pipe (the forward's time is its
// VALU-only time + its MFMA-only time, profiles/r05_mlp_rc_fwd_ablations.txt), so the kernels are priced in VALU CYCLES: see the
// activation helpers below.
#include "common.h"
#include <type_traits>
#ifndef RC_ABL
#define RC_ABL 0        // development ablations of mlp_rc_fwd3_kernel (tools/build_variant.py): 1 no activation, 2 no product 1, 3 no product 2, 4 no MFMA, 6 dropout only
#endif

typedef float rc_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 rc_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned rc_u4 __attribute__((ext_vector_type(4)));
typedef unsigned rc_u2 __attribute__((ext_vector_type(2)));
typedef float rc_f4 __attribute__((ext_vector_type(4)));
typedef short rc_v4i16 __attribute__((ext_vector_type(4)));
typedef short rc_v8i16 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) rc_v4i16* rc_lds_v4i16_ptr;

namespace {

// 16-byte-chunk XOR swizzle of a [rows][ROWB bytes] bf16 tile: conflict-free row-per-lane ds_read_b128 (lane groups of MI355X_MICROARCH
// "LDS") and, for 128-byte rows, at most 2-way ds_read_b64_tr_b16 of 4-row blocks
template <int ROWB>
__device__ __forceinline__ int rc_swz(int row) {
    if (ROWB == 64) return (row >> 2) & 3;
    if (ROWB == 128) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return row & 15;
}

// one 1 KiB piece of a [rows][ROWB] plane: HBM/L2 -> LDS by global_load_lds (the LDS side is wave-uniform base + lane * 16: the swizzle goes
// on the SOURCE address).  src: bf16 plane, element (row, k) at src[row * ld + k].
template <int ROWB>
__device__ __forceinline__ void rc_glds_piece(const uint16_t* __restrict__ src, long ld, int piece, int lane, char* tile) {
    constexpr int LPR = ROWB / 16, RPP = 1024 / ROWB;          // lanes per row, rows per piece
    const int row = piece * RPP + lane / LPR;
    const int lc = (lane % LPR) ^ rc_swz<ROWB>(row);
    __builtin_amdgcn_global_load_lds(src + (long)row * ld + (lc << 3), (__attribute__((address_space(3))) void*)(tile + piece * 1024), 16, 0, 0);
}

// MFMA operand fragment of a lane: 8 consecutive k (one 16-byte chunk) of `row`
template <int ROWB>
__device__ __forceinline__ rc_bf16x8 rc_frag(const char* tile, int row, int chunk) {
    return __builtin_bit_cast(rc_bf16x8, *reinterpret_cast<const rc_u4*>(tile + row * ROWB + ((chunk ^ rc_swz<ROWB>(row)) << 4)));
}

// The k-permuted operand fragment of a [rows][32 k] tile (64-byte rows) for the CHAINED products: product 1's accumulator leaves lane (token, g = lane >> 5) with the
// hidden units 4 g + {0..3} + 8 j (j = register quad); taken as they are, the registers of a 16-wide half hf are the k slots 0..3 <- k = 16 hf + 4 g ..+3 and
// 4..7 <- k = 16 hf + 8 + 4 g ..+3 -- so the WEIGHT operand is read in that order (two 8-byte reads) and the chained operand needs no lane exchange (8
// v_permlane32_swap per hidden step at ~8 cycles each; the two ds_read_b64 are 2-way bank conflicted, LDS is idle here).  (Round 3 put the values into the natural k
// order instead: the GEMM's summation order bit for bit; this order differs from it in the last bits.)
__device__ __forceinline__ rc_bf16x8 rc_frag_perm(const char* tile, int row, int hf, int g) {
    const int f = rc_swz<64>(row);
    // (volatile: hipcc otherwise pairs the reads of DIFFERENT fragments -- hi / lo plane, the two 32-row blocks -- into ds_read2st64_b64 and pays 24 v_mov per hidden step
    //  to put the operand quads back together)
    typedef __attribute__((address_space(3))) const volatile rc_u2* lds_u2_ptr;
    const rc_u2 a = *(lds_u2_ptr)(tile + row * 64 + (((2 * hf) ^ f) << 4) + 8 * g);
    const rc_u2 b = *(lds_u2_ptr)(tile + row * 64 + (((2 * hf + 1) ^ f) << 4) + 8 * g);
    return __builtin_bit_cast(rc_bf16x8, (rc_u4{a[0], a[1], b[0], b[1]}));
}

// P1 (round 5, the bf16 speed mode: mdvit_mlp_rc_planes(1)): ONE bf16 plane per operand -- the hi x hi product alone, a third of the MFMA work and no lo split of the chained
// operand; the lo planes still travel through the rings (L2 -> LDS is not what binds these kernels) and their fragment reads are dead code
#define RC_MFMA3P(P1, acc, ah, al, bh, bl)                                          \
    do {                                                                            \
        if constexpr (!(P1)) {                                                      \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);    \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);    \
        }                                                                           \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);        \
    } while (0)
#define RC_MFMA3(acc, ah, al, bh, bl)                                           \
    do {                                                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);    \
    } while (0)
// D[i][j] += sum_k A[i][k] B[k][j]: A-fragment lane (i = l & 31, k = 8 (l >> 5) ..+7), B-fragment lane (j = l & 31, same k); D: lane j,
// registers i = (r & 3) + 8 (r >> 2) + 4 (l >> 5).  Order lo*hi, hi*lo, hi*hi with the WEIGHT as the lo factor first: gemm_body.inc's.

// fp32 x8 -> hi / lo bf16 x8 (registers 0..3 <-> first quad, 4..7 <-> second quad)
__device__ __forceinline__ void rc_split8(const float* v, rc_u4& hi, rc_u4& lo) {
    uint2 h0, l0, h1, l1;
    mdvit_split_bf16x3(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
    mdvit_split_bf16x3(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
    hi = rc_u4{h0.x, h0.y, h1.x, h1.y};
    lo = rc_u4{l0.x, l0.y, l1.x, l1.y};
}

// A lane of product 1's accumulator holds, for the 16-wide hidden block b, the values k = 4 lhi + j (quad 2b) and k = 8 + 4 lhi + j (quad 2b + 1).
// The natural operand order wants k = 8 lhi .. 8 lhi + 7 in one lane: lanes l < 32 hand their second quad to l + 32 and take its first.
__device__ __forceinline__ rc_bf16x8 rc_natural_order(rc_u4 v) {
    const rc_u2 a = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
    const rc_u2 b = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
    return __builtin_bit_cast(rc_bf16x8, (rc_u4{a[0], b[0], a[1], b[1]}));
}

// the x / gm operand fragments of a wave's 32 tokens (rows past M re-read the last row: they only feed outputs that are never stored)
// LayerNorm prologue parameters: x is the LayerNorm's INPUT; the kernel normalises its rows in registers, writes the statistics and the normalised rows
// (operands of the backward kernels) and multiplies those
struct LnPro { const float* g; const float* b; float* mean; float* rstd; float* out; float eps; int rows_per_group; };
template <int K>
__device__ __forceinline__ void rc_ln_rows32(float4 (&a)[K / 16], float4 (&b)[K / 16], int row, int rowc, int M, int lhi, const LnPro ln);

template <int C, bool LNP = false>
__device__ __forceinline__ void rc_load_rows(const float* __restrict__ src, int row, int M, int lhi, rc_bf16x8 (&hi)[C / 16], rc_bf16x8 (&lo)[C / 16],
                                             const LnPro* ln = nullptr) {
    const float* p = src + (long)min(row, M - 1) * C + 8 * lhi;
    float4 a[C / 16], b[C / 16];
#pragma unroll
    for (int kb = 0; kb < C / 16; ++kb) {
        a[kb] = *reinterpret_cast<const float4*>(p + 16 * kb);
        b[kb] = *reinterpret_cast<const float4*>(p + 16 * kb + 4);
    }
    if constexpr (LNP) rc_ln_rows32<C>(a, b, row, min(row, M - 1), M, lhi, *ln);
#pragma unroll
    for (int kb = 0; kb < C / 16; ++kb) {
        const float v[8] = {a[kb].x, a[kb].y, a[kb].z, a[kb].w, b[kb].x, b[kb].y, b[kb].z, b[kb].w};
        rc_u4 h, l;
        rc_split8(v, h, l);
        hi[kb] = __builtin_bit_cast(rc_bf16x8, h);
        lo[kb] = __builtin_bit_cast(rc_bf16x8, l);
    }
}

// ---- the activation of the fused kernels (round 5) -----------------------------------------------------------------------------------------
// What an instruction costs here (tools/probe/valu_rates.hip, profiles/r05_valu_rates.txt; cycles per wave-instruction per SIMD, >= 2 waves per SIMD): f32 mul / add / fma and
// VOP2 integer logic ~2.5, everything VOP3-only or SDWA / DPP (v_bfi, v_alignbit, v_cvt_pk_bf16, v_cmp + v_cndmask, v_mul_lo_u32, v_pk_*_f32) ~4.5, v_exp / v_rcp and
// v_permlane32_swap ~8.2 -- so v_pk_fma_f32 buys nothing (two elements for the price of two v_fma), and an MFMA does NOT run beside a saturated VALU pipe of the same SIMD
// (tools/probe/mfma_valu_overlap.hip, profiles/r05_mfma_valu_overlap.txt; profiles/r05_mlp_rc_fwd_ablations.txt: the forward's time = its VALU-only time + its MFMA-only
// time): every VALU cycle removed is a cycle off the kernel.  Hence: no copysign (v_bfi), no separate dropout multiply, no re-derived indices.
//
// erf by Abramowitz-Stegun 7.1.26 (common.h's gelu_parts): q = erf(|x| / sqrt 2) = 1 - P(t) t e, t = 1 / (1 + p |x|), e = exp(-x^2 / 2) = exp2(x * x * (-0.5 log2 e)).
__device__ __forceinline__ void rc_erf_parts(float x, float& e, float& q) {
    e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    q = fmaf(-(poly * t), e, 1.0f);
}
// forward: hk = half the dropout keep-scale of the element (0.5 without dropout, 0.5 / keep kept, 0 dropped).  2 hk x Phi(x) with Phi(x) = 0.5 (1 + sign(x) q):
// hx + |hx| q, hx = hk x -- 13 instructions (2 transcendental) instead of 16 + the dropout multiply; the rounding of x Phi(x) moves by an ulp against the GEMM epilogue's form
__device__ __forceinline__ float rc_gelu_k(float x, float hk) {
    float e, q;
    rc_erf_parts(x, e, q);
    const float hx = x * hk;
    return fmaf(fabsf(hx), q, hx);
}
// backward: d gelu'(x) ks with gelu'(x) = Phi(x) + x phi(x), the keep-scale ks folded into the constants (hks = 0.5 ks, pks = ks / sqrt(2 pi)); d = 0 for a dropped element
__device__ __forceinline__ float rc_dgelu_k(float x, float d, float hks, float pks) {
    float e, q;
    rc_erf_parts(x, e, q);
    const float cdf = fmaf(copysignf(q, x), hks, hks);
    return d * fmaf(x * e, pks, cdf);
}
// the weight-gradient kernel wants both from one evaluation: h = 2 hk x Phi(x) (hk as rc_gelu_k) and du = d gelu'(x) ks (d, hks, pks as rc_dgelu_k)
__device__ __forceinline__ void rc_gelu_both_k(float x, float hk, float d, float hks, float pks, float& h, float& du) {
    float e, q;
    rc_erf_parts(x, e, q);
    const float hx = x * hk;
    h = fmaf(fabsf(hx), q, hx);
    const float cdf = fmaf(copysignf(q, x), hks, hks);
    du = d * fmaf(x * e, pks, cdf);
}
typedef float rc_f2 __attribute__((ext_vector_type(2)));
// hi / lo bf16 planes of a pair of values: mdvit_split_bf16x3's arithmetic (RNE hi, RNE (x - hi))
template <bool P1 = false>
__device__ __forceinline__ void rc_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector((rc_f2{a, b}), mdvit_bf16x2));
    if constexpr (P1) { lo = 0u; return; }          // one plane: the lo half is never multiplied
    const rc_f2 l = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(l, mdvit_bf16x2));
}
// The keep decisions of the four elements of an aligned group from the group's hash word (mdvit_drop_keep4), as select results: on / off per element
__device__ __forceinline__ void rc_keep_sel4(uint32_t h, uint32_t thresh16, float on, float off, float (&k)[4]) {
    bool keep[4];
    mdvit_drop_keep4(h, thresh16, keep);
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = keep[j] ? on : off;
}

// sum over the 16 channel quads of a row in the order of norm.hip's sum16 (xor 8, 4, 2, 1 over the quad index): a lane of the 32-token layout holds the
// quads 4 kq + 2 lhi + {0, 1} -- the steps over bit 3 and bit 2 (kq) are in-lane, bit 1 (lhi) is the partner lane, bit 0 is in-lane.  q[kq][b]: per-quad
// sums (C = 128: quads Q and Q + 16 of ln_fwd16's lane are kb and kb + 4 here, already added in ITS order by the caller).
__device__ __forceinline__ float lin_sum16(const float (&q)[4][2]) {
    const float p00 = q[0][0] + q[2][0], p01 = q[0][1] + q[2][1], p10 = q[1][0] + q[3][0], p11 = q[1][1] + q[3][1];     // xor 8
    float r0 = p00 + p10, r1 = p01 + p11;                                                                             // xor 4
    r0 += __shfl_xor(r0, 32, 64); r1 += __shfl_xor(r1, 32, 64);                                                        // xor 2
    return r0 + r1;                                                                                                    // xor 1
}

// In place on the raw rows of the 32-token layout (lane (token l & 31, lhi = l >> 5): a[kb] = channels 16 kb + 8 lhi .. +3, b[kb] = .. +4 .. +7):
// ln_fwd16_kernel's arithmetic, sum for sum (norm.hip) -- mean, then the centred squares, both through the same 16-quad tree; C = K in {64, 128}
template <int K>
__device__ __forceinline__ void rc_ln_rows32(float4 (&a)[K / 16], float4 (&b)[K / 16], int row, int rowc, int M, int lhi, const LnPro ln) {
    constexpr int KB = K / 16, VPL = K / 64;
    const int grp = rowc / ln.rows_per_group;
    const float* gg = ln.g + (long)grp * K + 8 * lhi;
    const float* gb = ln.b + (long)grp * K + 8 * lhi;
    float q[4][2];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {          // (C = 128: quad Q + 16 of ln_fwd16's lane <-> kb + 4 here)
            const float4 va = a[kq + 4 * j], vb = b[kq + 4 * j];
            s0 += (va.x + va.y) + (va.z + va.w);
            s1 += (vb.x + vb.y) + (vb.z + vb.w);
        }
        q[kq][0] = s0; q[kq][1] = s1;
    }
    const float mu = lin_sum16(q) * (1.0f / K);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        a[kb].x -= mu; a[kb].y -= mu; a[kb].z -= mu; a[kb].w -= mu;
        b[kb].x -= mu; b[kb].y -= mu; b[kb].z -= mu; b[kb].w -= mu;
    }
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            s0 += mdvit_ln_sq4(a[kq + 4 * j]);
            s1 += mdvit_ln_sq4(b[kq + 4 * j]);
        }
        q[kq][0] = s0; q[kq][1] = s1;
    }
    const float rs = mdvit_ln_rstd(lin_sum16(q), 1.0f / K, ln.eps);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float4 ga = *reinterpret_cast<const float4*>(gg + 16 * kb), gb4 = *reinterpret_cast<const float4*>(gg + 16 * kb + 4);
        const float4 ba = *reinterpret_cast<const float4*>(gb + 16 * kb), bb4 = *reinterpret_cast<const float4*>(gb + 16 * kb + 4);
        a[kb] = mdvit_ln_affine4(a[kb], rs, ga, ba);
        b[kb] = mdvit_ln_affine4(b[kb], rs, gb4, bb4);
        if (row < M) {
            *reinterpret_cast<float4*>(ln.out + (long)row * K + 8 * lhi + 16 * kb) = a[kb];
            *reinterpret_cast<float4*>(ln.out + (long)row * K + 8 * lhi + 16 * kb + 4) = b[kb];
        }
    }
    if (lhi == 0 && row < M) { ln.mean[row] = mu; ln.rstd[row] = rs; }
}

struct RcArgs {
    const float* x; const float* gm; const float* res; const float* rowscale; const float* b1; const float* b2;
    const uint16_t* W1p;    // planes of W1   [2][Hd][C]   rows = hidden, k = c         u = x W1^T
    const uint16_t* W2p;    // planes of W2   [2][C][Hd]   rows = c_out,  k = hidden    y = h W2^T
    const uint16_t* W2tp;   // planes of W2^T [2][Hd][C]   rows = hidden, k = c         d = gm W2
    const uint16_t* W1tp;   // planes of W1^T [2][C][Hd]   rows = c,      k = hidden    dx = du W1
    float* y; float* dx;
    float* du;              // rc16 dgrad: optional [tokens, hidden] copy of the hidden-layer gradient (the two weight-gradient GEMMs' operand)
    int hbf;                // h / du are stored as bf16 (2-byte elements behind the float pointers)
    float* h;               // rc16 forward: optional [tokens, hidden] copy of drop1(gelu(u)) (the fc2 weight-gradient GEMM's operand)
    LnPro ln;               // forward kernels with the LayerNorm prologue (LNP): x is the LayerNorm's input
    float* part;            // wgrad: per-workgroup partial sums
    int M, Hd, rows_per_scale;
    int drop; uint32_t k1a, k1b, k2a, k2b, thresh; float inv_keep;
    const uint32_t* seed;
    int groups, tiles_per_group;       // wgrad: token groups, 32-token tiles per group
};

// Scheduling request for a straight-line loop body of NM MFMAs, ND LDS reads and the VALU work of an activation: LEAD reads first, then
// 1 MFMA : V VALU : 1 read -- the operand reads run LEAD MFMAs ahead of their consumers (hipcc otherwise puts each ds_read + s_waitcnt right in
// front of its MFMA: ~100 exposed cycles per read on an in-order wave) and the activation's VALU issues underneath the matrix pipe.
template <int NM, int V, int ND, int LEAD>
__device__ __forceinline__ void rc_interleave() {
    __builtin_amdgcn_sched_group_barrier(0x100, LEAD, 0);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
        if (i < ND - LEAD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
}

#define RC_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// The ring barriers.  A slot is re-filled (LDS-DMA, issued right behind a barrier) while other waves may still have ds_reads of that slot QUEUED: the reads
// were issued before the barrier, but nothing orders an LDS-DMA write behind another wave's pending ds_read, and hipcc sinks the reads' s_waitcnt lgkmcnt
// below the barrier (to the first MFMA that uses them).  With the LDS pipeline of the CU congested -- a co-resident kernel of another stream issuing LDS
// atomics, e.g. the depthwise-convolution weight gradients next to the first C = 64 block backward of the sweep -- the DMA overtook the last reads of a phase
// (the lo-plane rows 32..63 of W1^T in mlp_rc_dgrad_kernel: single waves came out ~1e-3 off, in a few steps out of ten).  So every wave drains ITS LDS reads
// before it arrives: the barrier then also means "nobody reads the slots that are re-filled behind it".
#define RC_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ------------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------------
template <int C, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_rc_fwd_kernel(RcArgs p) {
    constexpr int KB = C / 16, CB = C / 32;
    constexpr int RB1 = C * 2;                       // row bytes of a W1 sub-tile [32 hidden][C]
    constexpr int T1 = 32 * RB1, T2 = C * 64;        // bytes of one plane of W1 sub [32][C] and W2 sub [C][32 hidden]
    constexpr int PIECES = (2 * T1 + 2 * T2) / 1024; // per hidden step: W1 hi, lo, W2 hi, lo
    static_assert(PIECES % NW == 0, "pieces must divide over the waves");
    constexpr int PPW = PIECES / NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* sW1 = smem;                                // [3 slots][2 planes][T1]
    char* sW2 = sW1 + 3 * 2 * T1;                    // [3 slots][2 planes][T2]
    float* sB1 = reinterpret_cast<float*>(sW2 + 3 * 2 * T2);      // [Hd]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    const int row = blockIdx.x * (NW * 32) + wave * 32 + l31;
    const int n = p.Hd >> 5;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1, k2a = p.k2a ^ s0, k2b = p.k2b + s1;
    const long wplane = (long)p.Hd * C;

    // piece pc of W1 sub-tile `src_s` -> ring slot `slot_s` % 3 (src_s != slot_s only for the clamped dummy fetches past the last step)
    auto issue_w1 = [&](int slot_s, int src_s, int pc) __attribute__((always_inline)) {
        constexpr int PP = T1 / 1024;
        const int pl = pc / PP, q = pc % PP;
        rc_glds_piece<RB1>(p.W1p + pl * wplane + (long)(src_s * 32) * C, C, q, lane, sW1 + ((slot_s % 3) * 2 + pl) * T1);
    };
    auto issue_w2 = [&](int s, int pc) __attribute__((always_inline)) {
        constexpr int PP = T2 / 1024;
        const int pl = pc / PP, q = pc % PP;
        rc_glds_piece<64>(p.W2p + pl * wplane + s * 32, p.Hd, q, lane, sW2 + ((s % 3) * 2 + pl) * T2);
    };
    // group g = {W1 sub (g + 2), W2 sub g}: PPW glds per wave, always (past the end the W1 part re-fetches the last sub-tile into the free slot,
    // so that the counted waits below see the same number of loads per group)
    auto issue_group = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave + i * NW;            // uniform per wave
            if (pc < 2 * T1 / 1024) issue_w1(g + 2, min(g + 2, n - 1), pc);
            else issue_w2(g, pc - 2 * T1 / 1024);
        }
    };

    // prologue: W1 sub 0, 1 (same piece split as a group), group 0, biases to LDS, x fragments
#pragma unroll
    for (int i = 0; i < 2 * T1 / 1024 / NW + (2 * T1 / 1024 % NW != 0); ++i) {
        const int pc = wave + i * NW;
        if (pc < 2 * T1 / 1024) { issue_w1(0, 0, pc); issue_w1(1, 1, pc); }
    }
    issue_group(0);
    for (int i = tid; i < p.Hd / 4; i += NW * 64) reinterpret_cast<float4*>(sB1)[i] = reinterpret_cast<const float4*>(p.b1)[i];
    rc_bf16x8 xh[KB], xl[KB];
    rc_load_rows<C>(p.x, row, p.M, lhi, xh, xl);
    rc_f32x16 yacc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[cb][r] = 0.f;
    RC_WAIT_VM(0);
    __syncthreads();

    auto prod1 = [&](int s, rc_f32x16& u) __attribute__((always_inline)) {
        const char* hi = sW1 + ((s % 3) * 2) * T1; const char* lo = hi + T1;
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const rc_bf16x8 ah = rc_frag<RB1>(hi, l31, 2 * kb + lhi), al = rc_frag<RB1>(lo, l31, 2 * kb + lhi);
            RC_MFMA3(u, ah, al, xh[kb], xl[kb]);
        }
    };
    auto prod2 = [&](int s, const rc_bf16x8 (&hh)[2], const rc_bf16x8 (&hl)[2]) __attribute__((always_inline)) {
        const char* hi = sW2 + ((s % 3) * 2) * T2; const char* lo = hi + T2;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const rc_bf16x8 ah = rc_frag<64>(hi, cb * 32 + l31, 2 * half + lhi), al = rc_frag<64>(lo, cb * 32 + l31, 2 * half + lhi);
                RC_MFMA3(yacc[cb], ah, al, hh[half], hl[half]);
            }
    };
    auto act = [&](int s, const rc_f32x16& u, rc_bf16x8 (&hh)[2], rc_bf16x8 (&hl)[2]) __attribute__((always_inline)) {
        uint32_t ph[8], pl[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int hd = s * 32 + 8 * q + 4 * lhi;
            const rc_f4 b4 = *reinterpret_cast<const rc_f4*>(sB1 + hd);          // (ext_vector load: a HIP float4 LDS read makes hipcc drain the glds ring)
            float hk[4] = {0.5f, 0.5f, 0.5f, 0.5f};
            if (DROP) rc_keep_sel4(mdvit_drop_bits(k1a, k1b, (uint32_t)((long)row * p.Hd + hd)), p.thresh >> 16, 0.5f * p.inv_keep, 0.f, hk);
            const float h0 = rc_gelu_k(u[4 * q + 0] + b4.x, hk[0]), h1 = rc_gelu_k(u[4 * q + 1] + b4.y, hk[1]);
            const float h2 = rc_gelu_k(u[4 * q + 2] + b4.z, hk[2]), h3 = rc_gelu_k(u[4 * q + 3] + b4.w, hk[3]);
            rc_split2(h0, h1, ph[2 * q], pl[2 * q]);
            rc_split2(h2, h3, ph[2 * q + 1], pl[2 * q + 1]);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            hh[half] = rc_natural_order(rc_u4{ph[4 * half], ph[4 * half + 1], ph[4 * half + 2], ph[4 * half + 3]});
            hl[half] = rc_natural_order(rc_u4{pl[4 * half], pl[4 * half + 1], pl[4 * half + 2], pl[4 * half + 3]});
        }
    };

    // Ring discipline.  Phase t runs product 1 of step t + 1 (reads W1 sub t + 1), product 2 of step t - 1 (reads W2 sub t - 1) and the
    // activation of step t.  Group g = {W1 sub g + 2, W2 sub g} goes into the slots phase g - 2 read last, so it is issued at the top of
    // phase g - 1, behind that phase's barrier; phase t needs group t - 1, the older of the two groups then in flight: vmcnt(PPW).
    rc_f32x16 ucur, unext;
    rc_bf16x8 hh[2], hl[2], gh[2], gl[2];
    prod1(0, ucur);
    {   // phase 0
        RC_BARRIER();
        issue_group(1);
        prod1(1, unext);
        act(0, ucur, hh, hl);
        ucur = unext;
    }
    for (int t = 1; t + 1 < n; ++t) {
        RC_WAIT_VM(PPW);
        RC_BARRIER();
        issue_group(t + 1);
        prod1(t + 1, unext);
        prod2(t - 1, hh, hl);
        act(t, ucur, gh, gl);
        rc_interleave<3 * KB + 6 * CB, DROP ? 17 : 14, 2 * KB + 4 * CB + 4, 8>();
        ucur = unext;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) { hh[h2] = gh[h2]; hl[h2] = gl[h2]; }
    }
    {   // phase n - 1
        RC_WAIT_VM(PPW);
        RC_BARRIER();
        prod2(n - 2, hh, hl);
        act(n - 1, ucur, hh, hl);
    }
    RC_WAIT_VM(0);
    RC_BARRIER();
    prod2(n - 1, hh, hl);

    // epilogue: + b2, dropout, DropPath scale, + residual (operands requested together)
    {
        const int rowc = min(row, p.M - 1);
        float4 b2q[CB][4], rq[CB][4];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = cb * 32 + 8 * q + 4 * lhi;
                b2q[cb][q] = *reinterpret_cast<const float4*>(p.b2 + col);
                rq[cb][q] = *reinterpret_cast<const float4*>(p.res + (long)rowc * C + col);
            }
        const float rsc = p.rowscale ? p.rowscale[rowc / p.rows_per_scale] : 1.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = cb * 32 + 8 * q + 4 * lhi;
                const float4 b4 = b2q[cb][q];
                float4 v = make_float4(yacc[cb][4 * q + 0] + b4.x, yacc[cb][4 * q + 1] + b4.y, yacc[cb][4 * q + 2] + b4.z, yacc[cb][4 * q + 3] + b4.w);
                if (DROP) {
                    const float4 ds = mdvit_drop_scale4(k2a, k2b, (uint32_t)((long)row * C + col), p.thresh, p.inv_keep);
                    v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
                }
                v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
                const float4 r4 = rq[cb][q];
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                if (row < p.M) *reinterpret_cast<float4*>(p.y + (long)row * C + col) = v;
            }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// forward, high-occupancy variant: no software pipelining inside the wave (one u accumulator, ~150 VGPRs -> 3 waves per SIMD, 3 workgroups
// per CU); the overlap of one wave's MFMAs with another's activation VALU comes from the third wave instead.  Group g = {W1 sub g, W2 sub g},
// issued two steps ahead into a three-slot ring.
// ------------------------------------------------------------------------------------------------------------------------------
template <int C, int NW, bool DROP, int OCC = 3, bool STORE = false, bool LNP = false, bool P1 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void mlp_rc_fwd3_kernel(RcArgs p) {
    constexpr int KB = C / 16, CB = C / 32;
    constexpr int RB1 = C * 2;
    constexpr int T1 = 32 * RB1, T2 = C * 64;
    static_assert(T1 == T2 && T1 / 1024 == NW, "one 1 KiB piece of each of the four planes (W1 hi, lo, W2 hi, lo) per wave and hidden step");
    constexpr int PPW = 4;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* sW1 = smem;
    char* sW2 = sW1 + 3 * 2 * T1;
    float* sB1 = reinterpret_cast<float*>(sW2 + 3 * 2 * T2);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    const int row = blockIdx.x * (NW * 32) + wave * 32 + l31;
    const int n = p.Hd >> 5;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1, k2a = p.k2a ^ s0, k2b = p.k2b + s1;
    const uint32_t dq0 = (uint32_t)((long)row * p.Hd + 4 * lhi) >> 2, thresh16 = p.thresh >> 16;      // dropout: this lane's first group index (Hd % 32 == 0)
    const float hki = 0.5f * p.inv_keep;
    const long wplane = (long)p.Hd * C;
    // Weight ring.  Group g = the four planes of hidden step g -> slot g % 3; wave w brings piece w of each plane.  A lane's SOURCE offset inside a sub-tile never changes
    // (rc_glds_piece's map: row, swizzled 16-byte chunk), the sub-tile's base is wave-uniform: base in SGPRs + one 32-bit lane offset per tile kind, nothing to compute per step.
    uint32_t voff1, voff2;
    {
        constexpr int LPR = RB1 / 16, RPP = 1024 / RB1;
        const int r1 = wave * RPP + lane / LPR;
        voff1 = (uint32_t)(r1 * C + (((lane % LPR) ^ rc_swz<RB1>(r1)) << 3)) * 2u;
        const int r2 = wave * 16 + (lane >> 2);
        voff2 = (uint32_t)(r2 * p.Hd + (((lane & 3) ^ rc_swz<64>(r2)) << 3)) * 2u;
    }
    const char* gW1 = reinterpret_cast<const char*>(p.W1p);
    const char* gW2 = reinterpret_cast<const char*>(p.W2p);
    auto issue_group = [&](int g, int slot) __attribute__((always_inline)) {
        const int gs = min(g, n - 1);
        const char* b1p = gW1 + (long)gs * (32 * C * 2);
        const char* b2p = gW2 + (long)gs * (32 * 2);
        typedef __attribute__((address_space(3))) void* lds_ptr;
        __builtin_amdgcn_global_load_lds(b1p + voff1, (lds_ptr)(sW1 + (slot * 2 + 0) * T1 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b1p + wplane * 2 + voff1, (lds_ptr)(sW1 + (slot * 2 + 1) * T1 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b2p + voff2, (lds_ptr)(sW2 + (slot * 2 + 0) * T2 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b2p + wplane * 2 + voff2, (lds_ptr)(sW2 + (slot * 2 + 1) * T2 + wave * 1024), 16, 0, 0);
    };
    issue_group(0, 0);
    issue_group(1, 1);
    for (int i = tid; i < p.Hd / 4; i += NW * 64) reinterpret_cast<float4*>(sB1)[i] = reinterpret_cast<const float4*>(p.b1)[i];
    rc_bf16x8 xh[KB], xl[KB];
    rc_load_rows<C, LNP>(p.x, row, p.M, lhi, xh, xl, &p.ln);
    rc_f32x16 yacc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[cb][r] = 0.f;
    __syncthreads();                                   // sB1 visible (the group waits below cover the weight tiles)

    // one hidden step; the ring slot is a compile-time constant (the loop below is unrolled by three), so every LDS address of a step is a loop-invariant register + an
    // immediate: no address arithmetic per step
    auto step = [&](auto slot_c, int t) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_c)::value;
        if (STORE && t > 0) RC_WAIT_VM(PPW + 4);       // group t landed (younger: group t + 1 and the four h stores of step t - 1)
        else RC_WAIT_VM(PPW);                          // group t landed (group t + 1 may still be in flight)
        RC_BARRIER();
        issue_group(t + 2, (slot + 2) % 3);            // into the slot step t - 1 read
        const char* w1h = sW1 + (slot * 2) * T1; const char* w1l = w1h + T1;
        const char* w2h = sW2 + (slot * 2) * T2; const char* w2l = w2h + T2;
        // u = bias + x W1s^T as D[hidden][token]: the accumulator starts from the bias quads
        rc_f32x16 u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const rc_f4 b4 = *reinterpret_cast<const rc_f4*>(sB1 + t * 32 + 8 * q + 4 * lhi);
            u[4 * q + 0] = b4.x; u[4 * q + 1] = b4.y; u[4 * q + 2] = b4.z; u[4 * q + 3] = b4.w;
        }
#if RC_ABL != 2 && RC_ABL != 4
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const rc_bf16x8 ah = rc_frag<RB1>(w1h, l31, 2 * kb + lhi), al = rc_frag<RB1>(w1l, l31, 2 * kb + lhi);
            RC_MFMA3P(P1, u, ah, al, xh[kb], xl[kb]);
        }
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] += __builtin_bit_cast(float, (uint32_t)xh[r & 3][0] << 16);        // ablation: no product 1
#endif
        uint32_t ph[8], pl[8];                          // hi / lo planes of the pairs (2 i, 2 i + 1) of this lane's 16 hidden values
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float hk[4] = {0.5f, 0.5f, 0.5f, 0.5f};
            if (DROP) rc_keep_sel4(mdvit_drop_bits_q(k1a, k1b, dq0 + t * 8 + 2 * q), thresh16, hki, 0.f, hk);
#if RC_ABL == 1
            const float h0 = u[4 * q + 0] * hk[0], h1 = u[4 * q + 1] * hk[1], h2 = u[4 * q + 2] * hk[2], h3 = u[4 * q + 3] * hk[3];      // ablation: no GELU
#else
            const float h0 = rc_gelu_k(u[4 * q + 0], hk[0]), h1 = rc_gelu_k(u[4 * q + 1], hk[1]), h2 = rc_gelu_k(u[4 * q + 2], hk[2]), h3 = rc_gelu_k(u[4 * q + 3], hk[3]);
#endif
            if (STORE) { if (row < p.M) *reinterpret_cast<float4*>(p.h + (long)row * p.Hd + t * 32 + 8 * q + 4 * lhi) = make_float4(h0, h1, h2, h3); }
            rc_split2<P1>(h0, h1, ph[2 * q], pl[2 * q]);
            rc_split2<P1>(h2, h3, ph[2 * q + 1], pl[2 * q + 1]);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const rc_bf16x8 hh = __builtin_bit_cast(rc_bf16x8, (rc_u4{ph[4 * half], ph[4 * half + 1], ph[4 * half + 2], ph[4 * half + 3]}));
            const rc_bf16x8 hl = __builtin_bit_cast(rc_bf16x8, (rc_u4{pl[4 * half], pl[4 * half + 1], pl[4 * half + 2], pl[4 * half + 3]}));
#if RC_ABL != 3 && RC_ABL != 4
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const rc_bf16x8 ah = rc_frag_perm(w2h, cb * 32 + l31, half, lhi);
                rc_bf16x8 al = ah;
                if constexpr (!P1) al = rc_frag_perm(w2l, cb * 32 + l31, half, lhi);
                RC_MFMA3P(P1, yacc[cb], ah, al, hh, hl);
            }
#else
            asm volatile("" ::"v"(hh), "v"(hl));       // ablation: no product 2
#endif
        }
    };
    for (int t = 0; t < n; t += 3) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < n) step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < n) step(std::integral_constant<int, 2>{}, t + 2);
    }
    RC_WAIT_VM(0);                                     // the clamped dummy fetches of the last two steps

    {
        const int rowc = min(row, p.M - 1);
        float4 b2q[CB][4], rq[CB][4];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = cb * 32 + 8 * q + 4 * lhi;
                b2q[cb][q] = *reinterpret_cast<const float4*>(p.b2 + col);
                rq[cb][q] = *reinterpret_cast<const float4*>(p.res + (long)rowc * C + col);
            }
        const float rsc = p.rowscale ? p.rowscale[rowc / p.rows_per_scale] : 1.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = cb * 32 + 8 * q + 4 * lhi;
                const float4 b4 = b2q[cb][q];
                float4 v = make_float4(yacc[cb][4 * q + 0] + b4.x, yacc[cb][4 * q + 1] + b4.y, yacc[cb][4 * q + 2] + b4.z, yacc[cb][4 * q + 3] + b4.w);
                if (DROP) {
                    const float4 ds = mdvit_drop_scale4(k2a, k2b, (uint32_t)((long)row * C + col), p.thresh, p.inv_keep);
                    v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
                }
                v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
                const float4 r4 = rq[cb][q];
                v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                if (row < p.M) *reinterpret_cast<float4*>(p.y + (long)row * C + col) = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward, data path:  dx = ((gm W2) * gelu'(x W1^T + b1) * mask1) W1      (gm = the masked upstream gradient, mdvit_colsum_f32)
// Same structure; per hidden step u = x W1s^T and d = gm W2s as D[hidden][token], du = d * gelu'(u + b1) * mask in registers,
// dx^T += W1s^T-tile * du (the chained operand).  Rings: W1 sub, W2^T sub (needed one step ahead), W1^T sub (one step behind).
// ------------------------------------------------------------------------------------------------------------------------------
template <int C, int NW, bool DROP, bool P1 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_rc_dgrad_kernel(RcArgs p) {
    constexpr int KB = C / 16, CB = C / 32;
    constexpr int RB1 = C * 2;
    constexpr int T1 = 32 * RB1, T3 = C * 64;        // one plane of W1 sub / W2^T sub [32][C], of W1^T sub [C][32 hidden]
    constexpr int PIECES = (4 * T1 + 2 * T3) / 1024;
    static_assert(PIECES % NW == 0, "pieces must divide over the waves");
    constexpr int PPW = PIECES / NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* sW1 = smem;                                // [3][2][T1]
    char* sW2t = sW1 + 3 * 2 * T1;                   // [3][2][T1]
    char* sW1t = sW2t + 3 * 2 * T1;                  // [3][2][T3]
    float* sB1 = reinterpret_cast<float*>(sW1t + 3 * 2 * T3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    const int row = blockIdx.x * (NW * 32) + wave * 32 + l31;
    const int n = p.Hd >> 5;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1;
    const uint32_t dq0 = (uint32_t)((long)row * p.Hd + 4 * lhi) >> 2, thresh16 = p.thresh >> 16;      // dropout: this lane's first group index (Hd % 32 == 0)
    const float hks = 0.5f * p.inv_keep, pks = 0.39894228040143268f * p.inv_keep;
    const long wplane = (long)p.Hd * C;

    // Weight rings (round 5: as in mlp_rc_fwd3_kernel): wave w brings piece w of each of the six planes of a hidden step -- W1 hi, lo, W2^T hi, lo (sub-tile [32 hidden][C]), W1^T hi,
    // lo (sub-tile [C][32 hidden]).  A lane's source offset inside a sub-tile never changes, the sub-tile's base is wave-uniform: nothing to compute or to branch on per step.
    static_assert(T1 / 1024 == NW && T3 / 1024 == NW, "one 1 KiB piece of each of the six planes per wave and hidden step");
    uint32_t voff1, voff3;
    {
        constexpr int LPR = RB1 / 16, RPP = 1024 / RB1;
        const int r1 = wave * RPP + lane / LPR;
        voff1 = (uint32_t)(r1 * C + (((lane % LPR) ^ rc_swz<RB1>(r1)) << 3)) * 2u;
        const int r3 = wave * 16 + (lane >> 2);
        voff3 = (uint32_t)(r3 * p.Hd + (((lane & 3) ^ rc_swz<64>(r3)) << 3)) * 2u;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue_a = [&](int slot_s, int src_s) __attribute__((always_inline)) {          // W1 / W2^T sub-tile src_s -> slot slot_s % 3
        const int sl = slot_s % 3;
        const char* b1p = reinterpret_cast<const char*>(p.W1p) + (long)src_s * (32 * C * 2);
        const char* b2p = reinterpret_cast<const char*>(p.W2tp) + (long)src_s * (32 * C * 2);
        __builtin_amdgcn_global_load_lds(b1p + voff1, (lds_ptr)(sW1 + (sl * 2 + 0) * T1 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b1p + wplane * 2 + voff1, (lds_ptr)(sW1 + (sl * 2 + 1) * T1 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b2p + voff1, (lds_ptr)(sW2t + (sl * 2 + 0) * T1 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b2p + wplane * 2 + voff1, (lds_ptr)(sW2t + (sl * 2 + 1) * T1 + wave * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int s) __attribute__((always_inline)) {                           // W1^T sub-tile s -> slot s % 3
        const int sl = s % 3;
        const char* b3p = reinterpret_cast<const char*>(p.W1tp) + (long)s * (32 * 2);
        __builtin_amdgcn_global_load_lds(b3p + voff3, (lds_ptr)(sW1t + (sl * 2 + 0) * T3 + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(b3p + wplane * 2 + voff3, (lds_ptr)(sW1t + (sl * 2 + 1) * T3 + wave * 1024), 16, 0, 0);
    };
    auto issue_group = [&](int g) __attribute__((always_inline)) {          // {W1 / W2^T sub (g + 2), W1^T sub g}: PPW = 6 loads per wave
        issue_a(g + 2, min(g + 2, n - 1));
        issue_b(g);
    };
    issue_a(0, 0);
    issue_a(1, 1);
    issue_group(0);
    for (int i = tid; i < p.Hd / 4; i += NW * 64) reinterpret_cast<float4*>(sB1)[i] = reinterpret_cast<const float4*>(p.b1)[i];
    rc_bf16x8 xh[KB], xl[KB], gh[KB], gl[KB];
    rc_load_rows<C>(p.x, row, p.M, lhi, xh, xl);
    rc_load_rows<C>(p.gm, row, p.M, lhi, gh, gl);
    rc_f32x16 dxacc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxacc[cb][r] = 0.f;
    RC_WAIT_VM(0);
    __syncthreads();

    auto prod1 = [&](int s, rc_f32x16& u, rc_f32x16& d) __attribute__((always_inline)) {
        const char* w1h = sW1 + ((s % 3) * 2) * T1; const char* w1l = w1h + T1;
        const char* w2h = sW2t + ((s % 3) * 2) * T1; const char* w2l = w2h + T1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {               // u starts from the bias quads of this step (one add per element less in the activation)
            const rc_f4 b4 = *reinterpret_cast<const rc_f4*>(sB1 + s * 32 + 8 * q + 4 * lhi);
            u[4 * q + 0] = b4.x; u[4 * q + 1] = b4.y; u[4 * q + 2] = b4.z; u[4 * q + 3] = b4.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const rc_bf16x8 ah = rc_frag<RB1>(w1h, l31, 2 * kb + lhi), al = rc_frag<RB1>(w1l, l31, 2 * kb + lhi);
            RC_MFMA3P(P1, u, ah, al, xh[kb], xl[kb]);
            const rc_bf16x8 ch = rc_frag<RB1>(w2h, l31, 2 * kb + lhi), cl = rc_frag<RB1>(w2l, l31, 2 * kb + lhi);
            RC_MFMA3P(P1, d, ch, cl, gh[kb], gl[kb]);
        }
    };
    auto prod3 = [&](int s, const rc_bf16x8 (&dh)[2], const rc_bf16x8 (&dl)[2]) __attribute__((always_inline)) {
        const char* hi = sW1t + ((s % 3) * 2) * T3; const char* lo = hi + T3;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const rc_bf16x8 ah = rc_frag_perm(hi, cb * 32 + l31, half, lhi);
                rc_bf16x8 al = ah;
                if constexpr (!P1) al = rc_frag_perm(lo, cb * 32 + l31, half, lhi);
                RC_MFMA3P(P1, dxacc[cb], ah, al, dh[half], dl[half]);
            }
    };
    auto act = [&](int s, const rc_f32x16& u, const rc_f32x16& d, rc_bf16x8 (&dh)[2], rc_bf16x8 (&dl)[2]) __attribute__((always_inline)) {
        uint32_t ph[8], pl[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dm[4] = {d[4 * q + 0], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
            if (DROP) {
                bool keep[4];
                mdvit_drop_keep4(mdvit_drop_bits_q(k1a, k1b, dq0 + s * 8 + 2 * q), thresh16, keep);
#pragma unroll
                for (int j = 0; j < 4; ++j) dm[j] = keep[j] ? dm[j] : 0.f;
            }
            const float g0 = rc_dgelu_k(u[4 * q + 0], dm[0], hks, pks), g1 = rc_dgelu_k(u[4 * q + 1], dm[1], hks, pks);
            const float g2 = rc_dgelu_k(u[4 * q + 2], dm[2], hks, pks), g3 = rc_dgelu_k(u[4 * q + 3], dm[3], hks, pks);
            rc_split2<P1>(g0, g1, ph[2 * q], pl[2 * q]);
            rc_split2<P1>(g2, g3, ph[2 * q + 1], pl[2 * q + 1]);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            dh[half] = __builtin_bit_cast(rc_bf16x8, (rc_u4{ph[4 * half], ph[4 * half + 1], ph[4 * half + 2], ph[4 * half + 3]}));
            dl[half] = __builtin_bit_cast(rc_bf16x8, (rc_u4{pl[4 * half], pl[4 * half + 1], pl[4 * half + 2], pl[4 * half + 3]}));
        }
    };

    // two accumulator sets and two fragment sets alternate roles phase by phase (the loop body holds TWO phases: no register rotation)
    rc_f32x16 uA, dA, uB, dB;
    rc_bf16x8 dh[2], dl[2], eh[2], el[2];
    auto phase = [&](int t, rc_f32x16& ucur, rc_f32x16& dcur, rc_f32x16& unext, rc_f32x16& dnext, const rc_bf16x8 (&ph)[2], const rc_bf16x8 (&pl)[2],
                     rc_bf16x8 (&oh)[2], rc_bf16x8 (&ol)[2]) __attribute__((always_inline)) {
        RC_WAIT_VM(PPW);
        RC_BARRIER();
        issue_group(t + 1);
        prod1(t + 1, unext, dnext);
        prod3(t - 1, ph, pl);
        act(t, ucur, dcur, oh, ol);
    };
    prod1(0, uA, dA);
    {
        RC_BARRIER();
        issue_group(1);
        prod1(1, uB, dB);
        act(0, uA, dA, dh, dl);
    }
    for (int t = 1; t + 1 < n; t += 2) {              // n is even: phases 1 .. n-2 come in pairs
        phase(t, uB, dB, uA, dA, dh, dl, eh, el);
        phase(t + 1, uA, dA, uB, dB, eh, el, dh, dl);
    }
    {
        RC_WAIT_VM(PPW);
        RC_BARRIER();
        prod3(n - 2, dh, dl);
        act(n - 1, uB, dB, eh, el);
    }
    RC_WAIT_VM(0);
    RC_BARRIER();
    prod3(n - 1, eh, el);

    if (row < p.M) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = cb * 32 + 8 * q + 4 * lhi;
                *reinterpret_cast<float4*>(p.dx + (long)row * C + col) = make_float4(dxacc[cb][4 * q + 0], dxacc[cb][4 * q + 1], dxacc[cb][4 * q + 2], dxacc[cb][4 * q + 3]);
            }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// backward, parameter path:  dW1 = du^T x,  db1 = colsum(du),  dW2 = gm^T h   with  u = x W1^T + b1,  h = gelu(u) mask1,
// du = (gm W2) gelu'(u) mask1  recomputed per 32-token tile -- the [tokens, hidden] operands of the two weight-gradient GEMMs never exist.
// A workgroup (8 waves) owns a 256-wide hidden range ("role") and a range of token tiles; wave w owns the 32 hidden units 32 w .. +31 of
// the role: its W1 / W2^T rows stay in REGISTERS as MFMA B-fragments for the whole kernel, its 32x64 blocks of dW1 and dW2 stay in the
// accumulators.  Per token tile the workgroup stages x and gm once (fp32 -> hi / lo planes in LDS, double buffered); a wave computes
// u, d as D[token][hidden] (lane <-> hidden), so h^T / du^T are already the k = token operand of the weight-gradient products, and reads
// the other operand -- x, gm with k = token -- transposed out of the SAME LDS image with ds_read_b64_tr_b16.  Partial sums per workgroup
// go to `part`, added in a fixed order by rc_reduce_kernel (deterministic).  Workgroups of one token range (all roles) sit on one XCD
// at adjacent dispatch slots: the second role reads x / gm out of that XCD's L2.
// ------------------------------------------------------------------------------------------------------------------------------
#define RC_MFMA3_ACT_AP(P1, acc, ah, al, bh, bl)                                    \
    do {                                                                            \
        if constexpr (!(P1)) {                                                      \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);    \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);    \
        }                                                                           \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);        \
    } while (0)
#define RC_MFMA3_ACT_A(acc, ah, al, bh, bl)                                     \
    do {                                                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);    \
    } while (0)

// every lane of a quad reads lane `src` of its quad (DPP quad_perm broadcast; src is a constant after unrolling)
__device__ __forceinline__ uint32_t rc_quad_bcast(uint32_t v, int src) {
    switch (src) {
        case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x00, 0xf, 0xf, false);
        case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x55, 0xf, 0xf, false);
        case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xAA, 0xf, 0xf, false);
        default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xFF, 0xf, 0xf, false);
    }
}

// 8 k (= tokens t0 .. t0+3 and t0+8 .. t0+11) of column `col` out of a [32 tokens][128-byte rows] plane: two transposing reads
__device__ __forceinline__ rc_bf16x8 rc_tr8(const char* plane, int t0, int col0, int l15) {
    // lane i of a 16-lane group hands in piece i = (token t0 + (i >> 2), columns col0 + 4 (i & 3) ..+3) and receives column col0 + i
    const int r0 = t0 + (l15 >> 2), r1 = r0 + 8, c = col0 + 4 * (l15 & 3);
    const int o0 = r0 * 128 + (((c >> 3) ^ rc_swz<128>(r0)) << 4) + (c & 4) * 2;
    const int o1 = r1 * 128 + (((c >> 3) ^ rc_swz<128>(r1)) << 4) + (c & 4) * 2;
    const rc_v4i16 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rc_lds_v4i16_ptr)(plane + o0));
    const rc_v4i16 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rc_lds_v4i16_ptr)(plane + o1));
    const rc_v8i16 r = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    return __builtin_bit_cast(rc_bf16x8, r);
}

// DX (round 5, mdvit_mlp_rc_bwd): the same kernel ALSO forms the data gradient dx = du W1 -- the whole MLP backward from one evaluation of u, d and the activation (the
// separate mlp_rc_dgrad_kernel recomputes both: 24 of its 36 MFMAs per hidden step and all of its VALU work are this kernel's over again).  du leaves the registers with
// lane <-> hidden unit, but dx contracts over hidden: every wave writes its du^T [32 hidden][32 tokens] (hi / lo planes) to LDS, and after a barrier wave (cblk, tblk) forms
// the 16 x 16 block dx^T[16 cblk ..][16 tblk ..] over ALL 256 hidden units of the role on v_mfma_f32_16x16x32_bf16 -- A = W1^T rows (the role's [64][256] slice, resident
// in LDS), B = du^T read back with ds_read_b64_tr_b16 (k = hidden down the rows) -- so no partial sums cross waves.  Roles (256-wide hidden ranges) write their own
// dx partial [role][tokens][C]; the consumer adds them (mdvit_layernorm_bwd's dy2, or rc_sum_parts_kernel).
#define RC16_MFMA3_BWDP(P1, acc, ah, al, bh, bl)                                    \
    do {                                                                            \
        if constexpr (!(P1)) {                                                      \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);    \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);    \
        }                                                                           \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);        \
    } while (0)
#define RC16_MFMA3_BWD(acc, ah, al, bh, bl)                                     \
    do {                                                                        \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);    \
    } while (0)
typedef float rc_acc4 __attribute__((ext_vector_type(4)));
constexpr int RC_WT_ROW = 528;                       // bytes per row of the W1^T slice in LDS: 256 hidden x 2 bytes + 16 (rows 4 banks apart)
constexpr int RC_BWD_LDS = 2 * 64 * RC_WT_ROW + 8 * 2 * 2048;

template <int C, bool DROP, bool DX = false, bool P1 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_rc_wgrad_kernel(RcArgs p) {
    constexpr int KB = C / 16, CB = C / 32;
    constexpr int TP = 32 * 128;                     // one plane of a [32 tokens][64 c] tile (C = 64: 128-byte rows)
    static_assert(C == 64, "token-tile staging below is written for C = 64");
    __shared__ __attribute__((aligned(1024))) char smem[2 * 4 * TP];      // [2 buffers][x hi, x lo, gm hi, gm lo]
    extern __shared__ __attribute__((aligned(1024))) char dsm[];          // DX: [2 planes][64 c][RC_WT_ROW] W1^T slice | [8 waves][2 planes][32 hidden][64 B] du^T
    char* sWt = dsm;
    char* sDu = dsm + 2 * 64 * RC_WT_ROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5, l15 = lane & 15;
    const int roles = p.Hd >> 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int role = idx % roles, group = (idx / roles) * 8 + xcd;
    const int hs = role * 256 + wave * 32;           // this wave's hidden units hs .. hs + 31; lane <-> hs + l31
    const int ntiles = (p.M + 31) >> 5;
    const int t_beg = min(group * p.tiles_per_group, ntiles), t_end = min(t_beg + p.tiles_per_group, ntiles);
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1;
    const long wplane = (long)p.Hd * C;

    // resident B-fragments: W1[hid][c] and W2^T[hid][c] rows of this lane's hidden unit
    rc_bf16x8 w1h[KB], w1l[KB], w2h[KB], w2l[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const long o = (long)(hs + l31) * C + 16 * kb + 8 * lhi;
        w1h[kb] = __builtin_bit_cast(rc_bf16x8, *reinterpret_cast<const rc_u4*>(p.W1p + o));
        w1l[kb] = __builtin_bit_cast(rc_bf16x8, *reinterpret_cast<const rc_u4*>(p.W1p + wplane + o));
        w2h[kb] = __builtin_bit_cast(rc_bf16x8, *reinterpret_cast<const rc_u4*>(p.W2tp + o));
        w2l[kb] = __builtin_bit_cast(rc_bf16x8, *reinterpret_cast<const rc_u4*>(p.W2tp + wplane + o));
    }
    const float bias = p.b1[hs + l31];
    rc_f32x16 aw1[CB], aw2[CB];                       // dW1[hid rows][c lanes], dW2[c rows][hid lanes]
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { aw1[cb][r] = 0.f; aw2[cb][r] = 0.f; }
    float db1 = 0.f;
    const uint32_t drot = 8u * (lane & 3), thresh16 = p.thresh >> 16;
    const float hks = 0.5f * p.inv_keep, pks = 0.39894228040143268f * p.inv_keep;

    // staging map: thread -> (token row tid >> 4, float4 column tid & 15)
    const int srow = tid >> 4, sc4 = tid & 15;
    const int soff = srow * 128 + (((sc4 >> 1) ^ rc_swz<128>(srow)) << 4) + (sc4 & 1) * 8;
    float4 rx, rg;
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        const int row = t * 32 + srow, rc = min(row, p.M - 1);
        rx = *reinterpret_cast<const float4*>(p.x + (long)rc * C + sc4 * 4);
        rg = *reinterpret_cast<const float4*>(p.gm + (long)rc * C + sc4 * 4);
        if (row >= p.M) rg = make_float4(0.f, 0.f, 0.f, 0.f);          // rows past the end contribute nothing
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        char* b = smem + buf * 4 * TP;
        uint2 hi, lo;
        mdvit_split_bf16x3(rx, hi, lo);
        *reinterpret_cast<rc_u2*>(b + soff) = rc_u2{hi.x, hi.y};
        *reinterpret_cast<rc_u2*>(b + TP + soff) = rc_u2{lo.x, lo.y};
        mdvit_split_bf16x3(rg, hi, lo);
        *reinterpret_cast<rc_u2*>(b + 2 * TP + soff) = rc_u2{hi.x, hi.y};
        *reinterpret_cast<rc_u2*>(b + 3 * TP + soff) = rc_u2{lo.x, lo.y};
    };

    if (DX) {                                        // W1^T[c][role's 256 hidden], both planes: 4096 16-byte chunks
        for (int i = tid; i < 2 * 64 * 32; i += 512) {
            const int pl = i >> 11, c = (i >> 5) & 63, ch = i & 31;
            const rc_u4 v = *reinterpret_cast<const rc_u4*>(p.W1tp + pl * wplane + (long)c * p.Hd + role * 256 + ch * 8);
            *reinterpret_cast<rc_u4*>(sWt + (pl * 64 + c) * RC_WT_ROW + ch * 16) = v;
        }
    }
    if (t_beg < t_end) {
        load_tile(t_beg);
        store_tile(0);
    }
    __syncthreads();
    for (int t = t_beg; t < t_end; ++t) {
        const int buf = (t - t_beg) & 1;
        const char* xb = smem + buf * 4 * TP;
        const char* xhi = xb; const char* xlo = xb + TP; const char* ghi = xb + 2 * TP; const char* glo = xb + 3 * TP;
        if (t + 1 < t_end) load_tile(t + 1);
        // u, d as D[token][hidden]
        rc_f32x16 u, d;
#pragma unroll
        for (int r = 0; r < 16; ++r) { u[r] = bias; d[r] = 0.f; }          // (lane <-> hidden unit: one bias value for all 16 token rows)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const rc_bf16x8 ah = rc_frag<128>(xhi, l31, 2 * kb + lhi), al = rc_frag<128>(xlo, l31, 2 * kb + lhi);
            RC_MFMA3_ACT_AP(P1, u, ah, al, w1h[kb], w1l[kb]);
            const rc_bf16x8 ch = rc_frag<128>(ghi, l31, 2 * kb + lhi), cl = rc_frag<128>(glo, l31, 2 * kb + lhi);
            RC_MFMA3_ACT_AP(P1, d, ch, cl, w2h[kb], w2l[kb]);
        }
        // per 16-token half: h, du in registers (lane <-> hidden unit, register r <-> token (r & 3) + 8 (r >> 2) + 4 lhi), then the
        // weight-gradient products over k = those 16 tokens in the order the registers hold them
        // dropout bits: the mask index of (token, hidden) is token * Hd + hidden and ONE hash serves the four hidden units of an aligned
        // group -- here four adjacent lanes.  Lane c = l & 3 of a quad hashes the tokens of registers 4c .. 4c+3; every register then takes
        // its word from lane r >> 2 of the quad (DPP quad broadcast) and rotates out the byte lane of its own hidden unit.
        uint32_t hb[4];
        if (DROP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tok = t * 32 + j + 8 * (lane & 3) + 4 * lhi;
                hb[j] = mdvit_drop_bits(k1a, k1b, (uint32_t)((long)tok * p.Hd + hs + l31));
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            rc_u4 hh, hl, dh, dl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {              // pairs of token rows (r, r + 1)
                const int r = 8 * ks + 2 * i;
                float hv[2], gv[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float hk = 0.5f, dm = d[r + e];
                    if (DROP) {
                        // the word of row r comes from lane r >> 2 of the quad; this lane's hidden unit is element lane & 3 of the group: the top half of rotr(word, 8 (lane & 3))
                        const uint32_t w = rc_quad_bcast(hb[(r + e) & 3], (r + e) >> 2);
                        const bool keep = (__builtin_amdgcn_alignbit(w, w, drot) >> 16) >= thresh16;
                        hk = keep ? hks : 0.f; dm = keep ? dm : 0.f;
                    }
                    rc_gelu_both_k(u[r + e], hk, dm, hks, pks, hv[e], gv[e]);
                    db1 += gv[e];
                }
                uint32_t a0, a1, a2, a3;
                rc_split2<P1>(hv[0], hv[1], a0, a1);
                rc_split2<P1>(gv[0], gv[1], a2, a3);
                hh[i] = a0; hl[i] = a1; dh[i] = a2; dl[i] = a3;
            }
            const rc_bf16x8 hhf = __builtin_bit_cast(rc_bf16x8, hh), hlf = __builtin_bit_cast(rc_bf16x8, hl);
            const rc_bf16x8 dhf = __builtin_bit_cast(rc_bf16x8, dh), dlf = __builtin_bit_cast(rc_bf16x8, dl);
            if (DX) {
                // du^T row of this lane's hidden unit: the packed pairs are tokens 16 ks + 4 lhi + {0..3} (dh[0], dh[1]) and 16 ks + 8 + 4 lhi + {0..3} (dh[2], dh[3]) -- two 8-byte
                // pieces per plane; 16-byte chunks XOR-swizzled by (row >> 1) & 3 (the 32 rows of a wave share their column: 2-way instead of 8-way bank conflicts)
                char* base = sDu + wave * 4096 + l31 * 64 + 8 * lhi;
                const int f = (l31 >> 1) & 3;
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    const int ch = (2 * ks + pc) ^ f;
                    *reinterpret_cast<rc_u2*>(base + (ch << 4)) = rc_u2{dh[2 * pc], dh[2 * pc + 1]};
                    *reinterpret_cast<rc_u2*>(base + 2048 + (ch << 4)) = rc_u2{dl[2 * pc], dl[2 * pc + 1]};
                }
            }
            const int t0 = 16 * ks + 4 * lhi;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int col0 = cb * 32 + 16 * ((lane >> 4) & 1);
                // dW1[hid][c] += du^T x : A = du^T (registers), B = x[k = token][c] (transposing read)
                const rc_bf16x8 bxh = rc_tr8(xhi, t0, col0, l15), bxl = rc_tr8(xlo, t0, col0, l15);
                RC_MFMA3_ACT_AP(P1, aw1[cb], dhf, dlf, bxh, bxl);
                // dW2[c][hid] += gm^T h : A = gm^T[c][k = token] (transposing read), B = h (registers)
                const rc_bf16x8 agh = rc_tr8(ghi, t0, col0, l15), agl = rc_tr8(glo, t0, col0, l15);
                RC_MFMA3_ACT_AP(P1, aw2[cb], agh, agl, hhf, hlf);
            }
        }
        if (DX) {
            __syncthreads();                            // every wave's du^T is in LDS
            const int cblk = wave & 3, tblk = wave >> 2, g4 = lane >> 4;
            rc_acc4 acc = {0.f, 0.f, 0.f, 0.f};
            // B piece of this lane inside a 4-row block: row r0 + (l15 >> 2), token columns 16 tblk + 4 (l15 & 3) ..+3; the transposing read returns column 16 tblk + l15
            const int pr = l15 >> 2, pcb = 32 * tblk + 8 * (l15 & 3);
            // (measured, profiles/r05_mlp_rc_bwd.txt: requesting the operands of step k + 1 / k + 2 ahead of step k's MFMAs does not help -- 830 against 800 us, more spills: the
            //  phase is bound by its LDS bytes, 32 KB per wave and tile, not by the round trips)
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {            // the 32 hidden units of wave k8 = one K = 32 step
                const char* wa = sWt + (cblk * 16 + l15) * RC_WT_ROW + (4 * k8 + g4) * 16;
                const rc_bf16x8 ah = __builtin_bit_cast(rc_bf16x8, *reinterpret_cast<const rc_u4*>(wa));
                const rc_bf16x8 al = __builtin_bit_cast(rc_bf16x8, *reinterpret_cast<const rc_u4*>(wa + 64 * RC_WT_ROW));
                const char* dplane = sDu + k8 * 4096;
                const int ra = 8 * g4 + pr, rb = ra + 4;
                const int oa = ra * 64 + (((pcb >> 4) ^ ((ra >> 1) & 3)) << 4) + (pcb & 15), ob = rb * 64 + (((pcb >> 4) ^ ((rb >> 1) & 3)) << 4) + (pcb & 15);
                const rc_v4i16 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rc_lds_v4i16_ptr)(dplane + oa)), h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rc_lds_v4i16_ptr)(dplane + ob));
                const rc_v4i16 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rc_lds_v4i16_ptr)(dplane + 2048 + oa)), l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((rc_lds_v4i16_ptr)(dplane + 2048 + ob));
                const rc_bf16x8 bh = __builtin_bit_cast(rc_bf16x8, (rc_v8i16{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]}));
                const rc_bf16x8 bl = __builtin_bit_cast(rc_bf16x8, (rc_v8i16{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]}));
                RC16_MFMA3_BWDP(P1, acc, ah, al, bh, bl);
            }
            const int tok = t * 32 + tblk * 16 + l15;    // D: lane <-> token column, registers <-> rows c = 16 cblk + 4 g4 + r
            if (tok < p.M) *reinterpret_cast<float4*>(p.dx + ((long)role * p.M + tok) * C + cblk * 16 + 4 * g4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        if (t + 1 < t_end) store_tile(buf ^ 1);
        __syncthreads();
    }

    // partial sums of this workgroup: part[group][ dW1 (Hd x C) | dW2 (C x Hd) | db1 (Hd) ]
    float* pg = p.part + (long)group * (2L * p.Hd * C + p.Hd);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            pg[(long)(hs + i) * C + cb * 32 + l31] = aw1[cb][r];
            pg[(long)p.Hd * C + (long)(cb * 32 + i) * p.Hd + hs + l31] = aw2[cb][r];
        }
    db1 += __shfl_xor(db1, 32, 64);
    if (lhi == 0) pg[2L * p.Hd * C + hs + l31] = db1;
}

// ==============================================================================================================================
// The data gradient on 16x16x32 MFMA tiles ("rc16"): a wave owns 16 tokens, so the x / gm operand fragments of C = 128 fit the register file
// (64 VGPRs) next to the 128 x 16 dx accumulator (32) -- the C = 128 stages' MLP backward data path (mpvit.py:71-78, hidden 1024) in ONE kernel:
//     u = x W1^T + b1 (recomputed), d = gm W2, du = d * gelu'(u) * mask1, dx += du W1          x, gm in -> dx (and optionally du) out
// instead of the recomputing fc2 data-gradient GEMM (du to HBM) + the fc1 data-gradient GEMM (du back from HBM).  du is written only when the
// weight-gradient GEMMs of the full sweep need it; the data-gradient-only sweep moves no [tokens, hidden] tensor at all.
//   v_mfma_f32_16x16x32_bf16: A lane (i = l & 15, k = 8 (l >> 4) ..+7), B lane (j = l & 15, same k), D lane j, registers i = 4 (l >> 4) + r.
// Products 1 / 2 run as D[hidden][token]: a lane holds the hidden units 4g .. 4g+3 (g = l >> 4) of each 16-row tile of ITS token; product 3
// contracts over the 32-wide hidden step with k slot (g, i) <-> hidden (i < 4 ? 4g + i : 16 + 4g + i - 4): the W1^T operand is READ in that
// order (two ds_read_b64 per fragment), so the chained operand needs no lane exchange.
// ==============================================================================================================================
typedef float rc_f32x4 __attribute__((ext_vector_type(4)));

template <int ROWB>
__device__ __forceinline__ int rc16_swz(int row) {       // lanes (row = l & 15 [+16], chunk = base + (l >> 4)): conflict-free ds_read_b128
    if (ROWB == 64) return (row >> 2) & 3;
    if (ROWB == 128) return (row >> 1) & 7;
    return row & 15;
}
template <int ROWB>
__device__ __forceinline__ void rc16_glds_piece(const uint16_t* __restrict__ src, long ld, int piece, int lane, char* tile) {
    constexpr int LPR = ROWB / 16, RPP = 1024 / ROWB;
    const int row = piece * RPP + lane / LPR;
    const int lc = (lane % LPR) ^ rc16_swz<ROWB>(row);
    __builtin_amdgcn_global_load_lds(src + (long)row * ld + (lc << 3), (__attribute__((address_space(3))) void*)(tile + piece * 1024), 16, 0, 0);
}
template <int ROWB>
__device__ __forceinline__ rc_bf16x8 rc16_frag(const char* tile, int row, int chunk) {
    return __builtin_bit_cast(rc_bf16x8, *reinterpret_cast<const rc_u4*>(tile + row * ROWB + ((chunk ^ rc16_swz<ROWB>(row)) << 4)));
}
// the permuted-k fragment of a [rows][32 k] tile (64-byte rows): k slots 0..3 <- k = 4g .. 4g+3, slots 4..7 <- k = 16 + 4g .. 16 + 4g + 3
__device__ __forceinline__ rc_bf16x8 rc16_frag_perm(const char* tile, int row, int g) {
    const int f = rc16_swz<64>(row);
    const rc_u2 a = *reinterpret_cast<const rc_u2*>(tile + row * 64 + (((g >> 1) ^ f) << 4) + (g & 1) * 8);
    const rc_u2 b = *reinterpret_cast<const rc_u2*>(tile + row * 64 + (((2 + (g >> 1)) ^ f) << 4) + (g & 1) * 8);
    return __builtin_bit_cast(rc_bf16x8, (rc_u4{a[0], a[1], b[0], b[1]}));
}

#define RC16_MFMA3P(P1, acc, ah, al, bh, bl)                                        \
    do {                                                                            \
        if constexpr (!(P1)) {                                                      \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);    \
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);    \
        }                                                                           \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);        \
    } while (0)
#define RC16_MFMA3(acc, ah, al, bh, bl)                                         \
    do {                                                                        \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);    \
    } while (0)

// x / gm operand fragments of a wave's 16 tokens: lane (token l & 15, g = l >> 4) holds c = 32 ks + 8 g ..+7 for every k step
__device__ __forceinline__ void rc16_ln_rows128(float4 (&a)[4], float4 (&b)[4], int row, int rowc, int M, int g, const LnPro ln);

template <int C, bool LNP = false>
__device__ __forceinline__ void rc16_load_rows(const float* __restrict__ src, int row, int M, int g, rc_bf16x8 (&hi)[C / 32], rc_bf16x8 (&lo)[C / 32],
                                               const LnPro* ln = nullptr) {
    const float* p = src + (long)min(row, M - 1) * C + 8 * g;
    float4 a[C / 32], b[C / 32];
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
        a[ks] = *reinterpret_cast<const float4*>(p + 32 * ks);
        b[ks] = *reinterpret_cast<const float4*>(p + 32 * ks + 4);
    }
    if constexpr (LNP && C == 128) rc16_ln_rows128(a, b, row, min(row, M - 1), M, g, *ln);
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
        const float v[8] = {a[ks].x, a[ks].y, a[ks].z, a[ks].w, b[ks].x, b[ks].y, b[ks].z, b[ks].w};
        rc_u4 h, l;
        rc_split8(v, h, l);
        hi[ks] = __builtin_bit_cast(rc_bf16x8, h);
        lo[ks] = __builtin_bit_cast(rc_bf16x8, l);
    }
}

// forward on the same tiles: y = res + rowscale * drop2( drop1(gelu(x W1^T + b1)) W2^T + b2 ), the hidden chunk chained in registers; STORE also
// writes h = drop1(gelu(u)) [tokens, hidden] once (C = 128: the fc2 weight-gradient GEMM reads it; the forward itself never re-reads it)
// The LayerNorm prologue in the 16-token layout, C = 128: lane (token l & 15, g = l >> 4) holds the channel quads 8 ks + 2 g + {0, 1} of the k steps ks = 0..3;
// ln_fwd16's lane `sub` holds the quads sub and sub + 16, i.e. ks and ks + 2 here; its tree over sub = 8 ks' + 2 g + b: bit 3 (ks') in-lane, bit 2 (g bit 1)
// lane ^ 32, bit 1 (g bit 0) lane ^ 16, bit 0 in-lane.
__device__ __forceinline__ float rc16_sum16(const float (&q)[2][2]) {
    float r0 = q[0][0] + q[1][0], r1 = q[0][1] + q[1][1];                          // xor 8
    r0 += __shfl_xor(r0, 32, 64); r1 += __shfl_xor(r1, 32, 64);                     // xor 4
    r0 += __shfl_xor(r0, 16, 64); r1 += __shfl_xor(r1, 16, 64);                     // xor 2
    return r0 + r1;                                                                 // xor 1
}
__device__ __forceinline__ void rc16_ln_rows128(float4 (&a)[4], float4 (&b)[4], int row, int rowc, int M, int g, const LnPro ln) {
    constexpr int K = 128;
    const int grp = rowc / ln.rows_per_group;
    const float* gg = ln.g + (long)grp * K + 8 * g;
    const float* gb = ln.b + (long)grp * K + 8 * g;
    float q[2][2];
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 va = a[kq + 2 * j], vb = b[kq + 2 * j];
            s0 += (va.x + va.y) + (va.z + va.w);
            s1 += (vb.x + vb.y) + (vb.z + vb.w);
        }
        q[kq][0] = s0; q[kq][1] = s1;
    }
    const float mu = rc16_sum16(q) * (1.0f / K);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a[ks].x -= mu; a[ks].y -= mu; a[ks].z -= mu; a[ks].w -= mu;
        b[ks].x -= mu; b[ks].y -= mu; b[ks].z -= mu; b[ks].w -= mu;
    }
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s0 += mdvit_ln_sq4(a[kq + 2 * j]);
            s1 += mdvit_ln_sq4(b[kq + 2 * j]);
        }
        q[kq][0] = s0; q[kq][1] = s1;
    }
    const float rs = mdvit_ln_rstd(rc16_sum16(q), 1.0f / K, ln.eps);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float4 ga = *reinterpret_cast<const float4*>(gg + 32 * ks), gb4 = *reinterpret_cast<const float4*>(gg + 32 * ks + 4);
        const float4 ba = *reinterpret_cast<const float4*>(gb + 32 * ks), bb4 = *reinterpret_cast<const float4*>(gb + 32 * ks + 4);
        a[ks] = mdvit_ln_affine4(a[ks], rs, ga, ba);
        b[ks] = mdvit_ln_affine4(b[ks], rs, gb4, bb4);
        if (row < M) {
            *reinterpret_cast<float4*>(ln.out + (long)row * K + 8 * g + 32 * ks) = a[ks];
            *reinterpret_cast<float4*>(ln.out + (long)row * K + 8 * g + 32 * ks + 4) = b[ks];
        }
    }
    if (g == 0 && row < M) { ln.mean[row] = mu; ln.rstd[row] = rs; }
}

template <int C, int NW, int OCC, bool DROP, bool STORE, bool LNP = false, bool P1 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void mlp_rc16_fwd_kernel(RcArgs p) {
    constexpr int KS = C / 32, CT = C / 16;
    constexpr int RB1 = C * 2;
    constexpr int T1 = 32 * RB1, T2 = C * 64;
    constexpr int PIECES = (2 * T1 + 2 * T2) / 1024;
    static_assert(PIECES % NW == 0, "pieces must divide over the waves");
    constexpr int PPW = PIECES / NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* sW1 = smem;
    char* sW2 = sW1 + 3 * 2 * T1;
    float* sB1 = reinterpret_cast<float*>(sW2 + 3 * 2 * T2);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, g = lane >> 4;
    const int row = blockIdx.x * (NW * 16) + wave * 16 + c16;
    const int n = p.Hd >> 5;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1, k2a = p.k2a ^ s0, k2b = p.k2b + s1;
    const uint32_t dq0 = (uint32_t)((long)row * p.Hd + 4 * g) >> 2, thresh16 = p.thresh >> 16;        // dropout: this lane's first group index (Hd % 32 == 0)
    const float hki = 0.5f * p.inv_keep;
    const long wplane = (long)p.Hd * C;
    auto issue_group = [&](int gi) __attribute__((always_inline)) {
        const int gs = min(gi, n - 1), slot = gi % 3;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave + i * NW;
            if (pc < 2 * T1 / 1024) {
                constexpr int PP = T1 / 1024;
                const int pl = pc / PP, q = pc % PP;
                rc16_glds_piece<RB1>(p.W1p + pl * wplane + (long)(gs * 32) * C, C, q, lane, sW1 + (slot * 2 + pl) * T1);
            } else {
                constexpr int PP = T2 / 1024;
                const int pc2 = pc - 2 * T1 / 1024, pl = pc2 / PP, q = pc2 % PP;
                rc16_glds_piece<64>(p.W2p + pl * wplane + gs * 32, p.Hd, q, lane, sW2 + (slot * 2 + pl) * T2);
            }
        }
    };
    issue_group(0);
    issue_group(1);
    for (int i = tid; i < p.Hd / 4; i += NW * 64) reinterpret_cast<float4*>(sB1)[i] = reinterpret_cast<const float4*>(p.b1)[i];
    rc_bf16x8 xh[KS], xl[KS];
    rc16_load_rows<C, LNP>(p.x, row, p.M, g, xh, xl, &p.ln);
    rc_f32x4 yacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) yacc[ct] = rc_f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int t = 0; t < n; ++t) {
        if (STORE && t > 0) RC_WAIT_VM(PPW + 2);      // (younger than group t: group t + 1 and the two h stores of step t - 1)
        else RC_WAIT_VM(PPW);
        RC_BARRIER();
        issue_group(t + 2);
        const int slot = t % 3;
        const char* w1h = sW1 + (slot * 2) * T1; const char* w1l = w1h + T1;
        const char* w2h = sW2 + (slot * 2) * T2; const char* w2l = w2h + T2;
        rc_u4 hh4, hl4;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const int hd = t * 32 + 16 * ht + 4 * g;              // this lane's four hidden units of the tile
            const rc_f4 b4 = *reinterpret_cast<const rc_f4*>(sB1 + hd);
            rc_f32x4 u = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const rc_bf16x8 ah = rc16_frag<RB1>(w1h, 16 * ht + c16, 4 * ks + g), al = rc16_frag<RB1>(w1l, 16 * ht + c16, 4 * ks + g);
                RC16_MFMA3P(P1, u, ah, al, xh[ks], xl[ks]);
            }
            float hk[4] = {0.5f, 0.5f, 0.5f, 0.5f};
            if (DROP) rc_keep_sel4(mdvit_drop_bits_q(k1a, k1b, dq0 + t * 8 + 4 * ht), thresh16, hki, 0.f, hk);
            const float h0 = rc_gelu_k(u[0], hk[0]), h1 = rc_gelu_k(u[1], hk[1]), h2 = rc_gelu_k(u[2], hk[2]), h3 = rc_gelu_k(u[3], hk[3]);
            if (STORE && !p.hbf) { if (row < p.M) *reinterpret_cast<float4*>(p.h + (long)row * p.Hd + hd) = make_float4(h0, h1, h2, h3); }
            uint32_t a0, a1, a2, a3;
            rc_split2<P1>(h0, h1, a0, a1);
            rc_split2<P1>(h2, h3, a2, a3);
            hh4[2 * ht] = a0; hl4[2 * ht] = a1; hh4[2 * ht + 1] = a2; hl4[2 * ht + 1] = a3;
        }
        if (STORE && p.hbf && row < p.M) {         // the hi plane IS bf16(h): two 8-byte stores (the two 4-unit groups of this lane) instead of two 16-byte ones
            uint16_t* hb = reinterpret_cast<uint16_t*>(p.h) + (long)row * p.Hd + t * 32 + 4 * g;
            *reinterpret_cast<uint2*>(hb) = make_uint2(hh4[0], hh4[1]);
            *reinterpret_cast<uint2*>(hb + 16) = make_uint2(hh4[2], hh4[3]);
        }
        const rc_bf16x8 hh = __builtin_bit_cast(rc_bf16x8, hh4), hl = __builtin_bit_cast(rc_bf16x8, hl4);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const rc_bf16x8 ah = rc16_frag_perm(w2h, 16 * ct + c16, g), al = rc16_frag_perm(w2l, 16 * ct + c16, g);
            RC16_MFMA3P(P1, yacc[ct], ah, al, hh, hl);
        }
    }
    RC_WAIT_VM(0);

    {   // epilogue: lane (token c16) holds output channels 16 ct + 4 g .. +3
        const int rowc = min(row, p.M - 1);
        float4 b2q[CT], rq[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = 16 * ct + 4 * g;
            b2q[ct] = *reinterpret_cast<const float4*>(p.b2 + col);
            rq[ct] = *reinterpret_cast<const float4*>(p.res + (long)rowc * C + col);
        }
        const float rsc = p.rowscale ? p.rowscale[rowc / p.rows_per_scale] : 1.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = 16 * ct + 4 * g;
            const float4 b4 = b2q[ct];
            float4 v = make_float4(yacc[ct][0] + b4.x, yacc[ct][1] + b4.y, yacc[ct][2] + b4.z, yacc[ct][3] + b4.w);
            if (DROP) {
                const float4 ds = mdvit_drop_scale4(k2a, k2b, (uint32_t)((long)row * C + col), p.thresh, p.inv_keep);
                v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
            }
            v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
            const float4 r4 = rq[ct];
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            if (row < p.M) *reinterpret_cast<float4*>(p.y + (long)row * C + col) = v;
        }
    }
}

template <int C, int NW, int OCC, bool DROP, bool STORE, bool P1 = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void mlp_rc16_dgrad_kernel(RcArgs p) {
    constexpr int KS = C / 32, CT = C / 16;
    constexpr int RB1 = C * 2;                       // row bytes of the [32 hidden][C] sub-tiles of W1 and W2^T
    constexpr int T1 = 32 * RB1, T3 = C * 64;        // bytes of one plane of a [32][C] sub-tile and of the W1^T sub-tile [C][32 hidden]
    constexpr int PIECES = (4 * T1 + 2 * T3) / 1024; // per hidden step: W1 hi, lo, W2^T hi, lo, W1^T hi, lo
    static_assert(PIECES % NW == 0, "pieces must divide over the waves");
    constexpr int PPW = PIECES / NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* sW1 = smem;                                // [3 slots][2 planes][T1]
    char* sW2 = sW1 + 3 * 2 * T1;                    // [3 slots][2 planes][T1]   (W2^T)
    char* sW3 = sW2 + 3 * 2 * T1;                    // [3 slots][2 planes][T3]   (W1^T)
    float* sB1 = reinterpret_cast<float*>(sW3 + 3 * 2 * T3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, g = lane >> 4;
    const int row = blockIdx.x * (NW * 16) + wave * 16 + c16;
    const int n = p.Hd >> 5;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k1a = p.k1a ^ s0, k1b = p.k1b + s1;
    const uint32_t dq0 = (uint32_t)((long)row * p.Hd + 4 * g) >> 2, thresh16 = p.thresh >> 16;
    const float hks = 0.5f * p.inv_keep, pks = 0.39894228040143268f * p.inv_keep;
    const long wplane = (long)p.Hd * C;
    // group gi = the three weight sub-tiles of hidden step gi (past the end: the last step again, into the free slot -- the counted waits
    // below then see the same number of loads per group)
    auto issue_group = [&](int gi) __attribute__((always_inline)) {
        const int gs = min(gi, n - 1), slot = gi % 3;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave + i * NW;            // uniform per wave
            constexpr int PP1 = T1 / 1024, PP3 = T3 / 1024;
            if (pc < 2 * PP1) {
                const int pl = pc / PP1, q = pc % PP1;
                rc16_glds_piece<RB1>(p.W1p + pl * wplane + (long)(gs * 32) * C, C, q, lane, sW1 + (slot * 2 + pl) * T1);
            } else if (pc < 4 * PP1) {
                const int pc2 = pc - 2 * PP1, pl = pc2 / PP1, q = pc2 % PP1;
                rc16_glds_piece<RB1>(p.W2tp + pl * wplane + (long)(gs * 32) * C, C, q, lane, sW2 + (slot * 2 + pl) * T1);
            } else {
                const int pc3 = pc - 4 * PP1, pl = pc3 / PP3, q = pc3 % PP3;
                rc16_glds_piece<64>(p.W1tp + pl * wplane + gs * 32, p.Hd, q, lane, sW3 + (slot * 2 + pl) * T3);
            }
        }
    };
    issue_group(0);
    issue_group(1);
    for (int i = tid; i < p.Hd / 4; i += NW * 64) reinterpret_cast<float4*>(sB1)[i] = reinterpret_cast<const float4*>(p.b1)[i];
    rc_bf16x8 xh[KS], xl[KS], mh[KS], ml[KS];
    rc16_load_rows<C>(p.x, row, p.M, g, xh, xl);
    rc16_load_rows<C>(p.gm, row, p.M, g, mh, ml);
    rc_f32x4 dacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dacc[ct] = rc_f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int t = 0; t < n; ++t) {
        // group t is the older of the two groups in flight; vector-memory operations retire in issue order, and younger than group t are group
        // t + 1 and (STORE) the two du stores of step t - 1
        if (STORE && t > 0) RC_WAIT_VM(PPW + 2);
        else RC_WAIT_VM(PPW);
        RC_BARRIER();
        issue_group(t + 2);
        const int slot = t % 3;
        const char* w1h = sW1 + (slot * 2) * T1; const char* w1l = w1h + T1;
        const char* w2h = sW2 + (slot * 2) * T1; const char* w2l = w2h + T1;
        const char* w3h = sW3 + (slot * 2) * T3; const char* w3l = w3h + T3;
        rc_u4 dh4, dl4;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const int hd = t * 32 + 16 * ht + 4 * g;              // this lane's four hidden units of the tile
            const rc_f4 b4 = *reinterpret_cast<const rc_f4*>(sB1 + hd);
            rc_f32x4 u = {b4.x, b4.y, b4.z, b4.w}, d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const rc_bf16x8 ah = rc16_frag<RB1>(w1h, 16 * ht + c16, 4 * ks + g), al = rc16_frag<RB1>(w1l, 16 * ht + c16, 4 * ks + g);
                RC16_MFMA3P(P1, u, ah, al, xh[ks], xl[ks]);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const rc_bf16x8 ah = rc16_frag<RB1>(w2h, 16 * ht + c16, 4 * ks + g), al = rc16_frag<RB1>(w2l, 16 * ht + c16, 4 * ks + g);
                RC16_MFMA3P(P1, d, ah, al, mh[ks], ml[ks]);
            }
            float dm[4] = {d[0], d[1], d[2], d[3]};
            if (DROP) {
                bool keep[4];
                mdvit_drop_keep4(mdvit_drop_bits_q(k1a, k1b, dq0 + t * 8 + 4 * ht), thresh16, keep);
#pragma unroll
                for (int j = 0; j < 4; ++j) dm[j] = keep[j] ? dm[j] : 0.f;
            }
            const float g0 = rc_dgelu_k(u[0], dm[0], hks, pks), g1 = rc_dgelu_k(u[1], dm[1], hks, pks), g2 = rc_dgelu_k(u[2], dm[2], hks, pks), g3 = rc_dgelu_k(u[3], dm[3], hks, pks);
            if (STORE && !p.hbf) { if (row < p.M) *reinterpret_cast<float4*>(p.du + (long)row * p.Hd + hd) = make_float4(g0, g1, g2, g3); }
            uint32_t a0, a1, a2, a3;
            rc_split2<P1>(g0, g1, a0, a1);
            rc_split2<P1>(g2, g3, a2, a3);
            dh4[2 * ht] = a0; dl4[2 * ht] = a1; dh4[2 * ht + 1] = a2; dl4[2 * ht + 1] = a3;
        }
        if (STORE && p.hbf && row < p.M) {
            uint16_t* db = reinterpret_cast<uint16_t*>(p.du) + (long)row * p.Hd + t * 32 + 4 * g;
            *reinterpret_cast<uint2*>(db) = make_uint2(dh4[0], dh4[1]);
            *reinterpret_cast<uint2*>(db + 16) = make_uint2(dh4[2], dh4[3]);
        }
        const rc_bf16x8 dh = __builtin_bit_cast(rc_bf16x8, dh4), dl = __builtin_bit_cast(rc_bf16x8, dl4);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const rc_bf16x8 ah = rc16_frag_perm(w3h, 16 * ct + c16, g), al = rc16_frag_perm(w3l, 16 * ct + c16, g);
            RC16_MFMA3P(P1, dacc[ct], ah, al, dh, dl);
        }
    }
    RC_WAIT_VM(0);
    if (row < p.M) {       // lane (token c16) holds the input channels 16 ct + 4 g .. +3
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            *reinterpret_cast<float4*>(p.dx + (long)row * C + 16 * ct + 4 * g) = make_float4(dacc[ct][0], dacc[ct][1], dacc[ct][2], dacc[ct][3]);
    }
}

// ==============================================================================================================================
// Streaming Linear for the short-K layers (K = 64 / 128: qkv and proj of the C = 64 / 128 stages, mdvit.py:288,310, their data gradients, and the
// 64 / 128 -> 512 projections of the peer heads, Decoders.py:320-331): y = x W^T (+ bias, dropout, DropPath scale, residual).  These products
// are HBM-bound and OUTPUT-heavy; the tiled GEMM (gemm.hip) runs them at 1.8-2.5 TB/s because every 64 / 128-row tile pays its own
// load -> LDS -> MFMA -> store latency chain.  Here a wave owns 32 tokens for the whole output row: x is loaded ONCE straight into MFMA operand
// registers (no LDS for activations), the weight arrives pre-split (the per-step bf16 planes) by global_load_lds through a three-slot ring,
// 32 output features per step, and every step ends in 16-byte stores -- the kernel is a stream of stores with the next weights in flight.
// Same bf16x3 arithmetic as the GEMM (lo*hi, hi*lo, hi*hi per k step, k ascending), same dropout keys / indices.
// ==============================================================================================================================
struct LinArgs {
    const float* x; long lda; const uint16_t* Wp; long wplane; const float* bias; float* y; long ldc;
    const float* residual; long ldr; const float* rowscale; int rows_per_scale;
    int M, N; uint32_t k0, k1, thresh; float inv_keep; const uint32_t* seed;
    // LN prologue (LNP): x is the LayerNorm's INPUT; the kernel normalises its 32 rows in registers, writes the statistics and the normalised rows
    // (the weight-gradient GEMM's operand) and multiplies those
    const float* ln_g; const float* ln_b; float* ln_mean; float* ln_rstd; float* ln_out; float ln_eps; int ln_rows_per_group;
};

template <int K, int NW, bool FULL, bool DROP, bool LNP = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(3, 4))) void lin_rc_kernel(LinArgs p) {
    constexpr int KB = K / 16;
    constexpr int RB = K * 2;                        // row bytes of a weight sub-tile [32 features][K]
    constexpr int T1 = 32 * RB;                      // bytes of one plane of it
    constexpr int PIECES = 2 * T1 / 1024;            // hi, lo
    static_assert(PIECES % NW == 0, "pieces must divide over the waves");
    constexpr int PPW = PIECES / NW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* sW = smem;                                 // [3 slots][2 planes][T1]
    float* sB = reinterpret_cast<float*>(sW + 3 * 2 * T1);      // [N] bias (zeros without one)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
    const int row = blockIdx.x * (NW * 32) + wave * 32 + l31, rowc = min(row, p.M - 1);
    const int n = p.N >> 5;
    uint32_t s0 = 0, s1 = 0;
    if (DROP && p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t k0 = p.k0 ^ s0, k1 = p.k1 + s1;
    auto issue_group = [&](int gi) __attribute__((always_inline)) {
        const int gs = min(gi, n - 1), slot = gi % 3;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave + i * NW;            // uniform per wave
            constexpr int PP = T1 / 1024;
            const int pl = pc / PP, q = pc % PP;
            rc_glds_piece<RB>(p.Wp + pl * p.wplane + (long)(gs * 32) * K, K, q, lane, sW + (slot * 2 + pl) * T1);
        }
    };
    issue_group(0);
    issue_group(1);
    for (int i = tid; i < p.N / 4; i += NW * 64)
        reinterpret_cast<float4*>(sB)[i] = p.bias ? reinterpret_cast<const float4*>(p.bias)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    rc_bf16x8 xh[KB], xl[KB];
    {
        const float* px = p.x + (long)rowc * p.lda + 8 * lhi;
        float4 a[KB], b[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            a[kb] = *reinterpret_cast<const float4*>(px + 16 * kb);
            b[kb] = *reinterpret_cast<const float4*>(px + 16 * kb + 4);
        }
        if (LNP) rc_ln_rows32<K>(a, b, row, rowc, p.M, lhi, LnPro{p.ln_g, p.ln_b, p.ln_mean, p.ln_rstd, p.ln_out, p.ln_eps, p.ln_rows_per_group});
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float v[8] = {a[kb].x, a[kb].y, a[kb].z, a[kb].w, b[kb].x, b[kb].y, b[kb].z, b[kb].w};
            rc_u4 h, l;
            rc_split8(v, h, l);
            xh[kb] = __builtin_bit_cast(rc_bf16x8, h);
            xl[kb] = __builtin_bit_cast(rc_bf16x8, l);
        }
    }
    const float rsc = (FULL && p.rowscale) ? p.rowscale[rowc / p.rows_per_scale] : 1.f;
    __syncthreads();

    for (int t = 0; t < n; ++t) {
        // the residual quads of this step are requested BEFORE the next weight group: the wait the compiler puts in front of their use then
        // leaves that group in flight
        float4 rq[4];
        if (FULL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rq[q] = *reinterpret_cast<const float4*>(p.residual + (long)rowc * p.ldr + t * 32 + 8 * q + 4 * lhi);
        }
        // weight group t must have landed.  Vector-memory operations retire in issue order; younger than group t are group t + 1 (PPW loads), the
        // four stores of step t - 1 and the four residual loads just requested
        if (t == 0) RC_WAIT_VM(PPW + (FULL ? 4 : 0));
        else RC_WAIT_VM(PPW + 4 + (FULL ? 4 : 0));
        RC_BARRIER();
        issue_group(t + 2);
        const char* hi = sW + ((t % 3) * 2) * T1; const char* lo = hi + T1;
        rc_f32x16 u;
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const rc_bf16x8 ah = rc_frag<RB>(hi, l31, 2 * kb + lhi), al = rc_frag<RB>(lo, l31, 2 * kb + lhi);
            RC_MFMA3(u, ah, al, xh[kb], xl[kb]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = t * 32 + 8 * q + 4 * lhi;
            const rc_f4 b4 = *reinterpret_cast<const rc_f4*>(sB + col);
            float4 v = make_float4(u[4 * q + 0] + b4.x, u[4 * q + 1] + b4.y, u[4 * q + 2] + b4.z, u[4 * q + 3] + b4.w);
            if (DROP) {
                const float4 ds = mdvit_drop_scale4(k0, k1, (uint32_t)((long)row * p.N + col), p.thresh, p.inv_keep);
                v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
            }
            if (FULL) {
                // (a multiply and an add, never fused: the tiled GEMM's FULL epilogue -- gemm_body.inc -- is the same arithmetic, bit for bit)
#pragma clang fp contract(off)
                v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
                v.x += rq[q].x; v.y += rq[q].y; v.z += rq[q].z; v.w += rq[q].w;
            }
            if (row < p.M) *reinterpret_cast<float4*>(p.y + (long)row * p.ldc + col) = v;
        }
    }
    RC_WAIT_VM(0);
}

// out_j[i] (+)= sum_g part[g][off_j + i] for the three segments, groups added in order: one float4 column per thread
__global__ __launch_bounds__(256) void rc_reduce_kernel(const float* __restrict__ part, int groups, long stride, int n0, float* o0, int n1, float* o1, int n2, float* o2,
                                                        int accumulate) {
    const int q = blockIdx.x * 256 + threadIdx.x, n = n0 + n1 + n2;
    if (q * 4 >= n) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = part + (long)q * 4;
    int g = 0;
    for (; g + 4 <= groups; g += 4) {
        const float4 a = *reinterpret_cast<const float4*>(src + (long)g * stride), b = *reinterpret_cast<const float4*>(src + (long)(g + 1) * stride);
        const float4 c = *reinterpret_cast<const float4*>(src + (long)(g + 2) * stride), d = *reinterpret_cast<const float4*>(src + (long)(g + 3) * stride);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; g < groups; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(src + (long)g * stride);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    const int i = q * 4;                             // n0, n1, n2 % 4 == 0: a quad never straddles two segments
    float* dst = i < n0 ? o0 + i : (i < n0 + n1 ? o1 + (i - n0) : o2 + (i - n0 - n1));
    float4 v = s;
    if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *reinterpret_cast<float4*>(dst) = v;
}

int rc_set_lds(const void* k, int bytes, bool (&flags)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (flags[dev]) return MDVIT_OK;
    const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "mlp_rc: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    flags[dev] = true;
    return MDVIT_OK;
}

void rc_fill(RcArgs& a, int M, int Hd, float drop_p, uint32_t k10, uint32_t k11, uint32_t k20, uint32_t k21, const uint32_t* seed) {
    a.M = M; a.Hd = Hd;
    a.drop = drop_p > 0.f; a.k1a = k10; a.k1b = k11; a.k2a = k20; a.k2b = k21;
    a.thresh = mdvit_drop_thresh(drop_p); a.inv_keep = 1.f / (1.f - drop_p);
    a.seed = seed;
}

}  // namespace

// One bf16 plane per operand in the register-chained MLP kernels (the bf16 speed mode) instead of the bf16x3 hi / lo pair: mdvit_mlp_rc_planes(1 | 2).  A process-wide
// arithmetic switch like ops.set_gemm_precision, which sets it; 2 (the parity arithmetic) unless told otherwise.
int g_rc_planes = 2;
extern "C" int mdvit_mlp_rc_planes(int32_t planes) {
    if (planes != 1 && planes != 2) return mdvit_set_error(MDVIT_E_SHAPE, "mlp_rc_planes: 1 or 2");
    g_rc_planes = planes;
    return MDVIT_OK;
}
// launch one instantiation: its dynamic-LDS limit raised once per device (lds_cap > 0), then the launch
template <auto KERNEL>
static int rc_launch(int lds_cap, dim3 grid, dim3 block, int smem, hipStream_t s, const RcArgs& a) {
    static bool fl[64] = {false};
    if (lds_cap > 0) {
        const int rc = rc_set_lds(reinterpret_cast<const void*>(KERNEL), lds_cap, fl);
        if (rc != MDVIT_OK) return rc;
    }
    hipLaunchKernelGGL(KERNEL, grid, block, smem, s, a);
    return MDVIT_OK;
}

int g_rc_fwd_variant = 3;       // 2: software-pipelined wave, 2 waves per SIMD; 3: plain wave, 3 waves per SIMD (tuning hook: mdvit_mlp_rc_config)
int g_rc_fwd128_variant = 16;   // C = 128 forward: 16-token waves (16x16x32 tiles; 327-352 us at 131072 tokens) or 32-token waves (32x32x16; 388-431 us: measured, not the default)
extern "C" int mdvit_mlp_rc_config(int32_t fwd_variant) {
    if (fwd_variant == 16 || fwd_variant == 32) g_rc_fwd128_variant = fwd_variant;
    else g_rc_fwd_variant = fwd_variant;
    return MDVIT_OK;
}

extern "C" int mdvit_mlp_rc_fwd(const float* x, const void* W1p, const float* b1, const void* W2p, const float* b2, const float* res, const float* rowscale,
                                int32_t rows_per_scale, float* y, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1,
                                uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(C == 64, MDVIT_E_SHAPE, "mlp_rc_fwd: built for C = 64 (got %d)", C);
    MDVIT_CHECK_ARG(M > 0 && Hd >= 64 && Hd % 64 == 0 && Hd <= 4096, MDVIT_E_SHAPE, "mlp_rc_fwd: need M > 0, hidden %% 64 == 0, hidden <= 4096 (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(x && W1p && b1 && W2p && b2 && res && y, MDVIT_E_SHAPE, "mlp_rc_fwd: null operand");
    MDVIT_CHECK_ARG(aligned16(x) && aligned16(W1p) && aligned16(b1) && aligned16(W2p) && aligned16(b2) && aligned16(res) && aligned16(y), MDVIT_E_ALIGN,
                    "mlp_rc_fwd: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp_rc_fwd: dropout index space exceeds 2^32");
    RcArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.W1p = (const uint16_t*)W1p; a.b1 = b1; a.W2p = (const uint16_t*)W2p; a.b2 = b2; a.res = res; a.rowscale = rowscale; a.y = y;
    a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    rc_fill(a, M, Hd, drop_p, key1_0, key1_1, key2_0, key2_1, drop_seed);
    constexpr int NW = 4;
    const int smem = 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + Hd * 4;
    static bool flags[64] = {false};
    static bool flags2[64] = {false};
    int rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_fwd_kernel<64, NW, false>), 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, flags);
    if (rc == MDVIT_OK) rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_fwd_kernel<64, NW, true>), 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, flags2);
    if (rc != MDVIT_OK) return rc;
    if (g_rc_fwd_variant == 3) {
        static bool f3[64] = {false}, f4[64] = {false};
        rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_fwd3_kernel<64, NW, false>), 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, f3);
        if (rc == MDVIT_OK) rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_fwd3_kernel<64, NW, true>), 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, f4);
        if (rc != MDVIT_OK) return rc;
        constexpr int CAP = 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4;
        const dim3 grid(cdiv(M, NW * 32)), block(NW * 64);
        hipStream_t s = (hipStream_t)stream;
        if (g_rc_planes == 1) rc = a.drop ? rc_launch<&mlp_rc_fwd3_kernel<64, NW, true, 3, false, false, true>>(CAP, grid, block, smem, s, a)
                                          : rc_launch<&mlp_rc_fwd3_kernel<64, NW, false, 3, false, false, true>>(CAP, grid, block, smem, s, a);
        else rc = a.drop ? rc_launch<&mlp_rc_fwd3_kernel<64, NW, true>>(CAP, grid, block, smem, s, a) : rc_launch<&mlp_rc_fwd3_kernel<64, NW, false>>(CAP, grid, block, smem, s, a);
        if (rc != MDVIT_OK) return rc;
    } else if (a.drop) hipLaunchKernelGGL((mlp_rc_fwd_kernel<64, NW, true>), dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_rc_fwd_kernel<64, NW, false>), dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_mlp_rc_dgrad(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* dx,
                                  int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(C == 64, MDVIT_E_SHAPE, "mlp_rc_dgrad: built for C = 64 (got %d)", C);
    MDVIT_CHECK_ARG(M > 0 && Hd >= 64 && Hd % 64 == 0 && Hd <= 4096, MDVIT_E_SHAPE, "mlp_rc_dgrad: need M > 0, hidden %% 64 == 0, hidden <= 4096 (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(gm && x && W1p && b1 && W2tp && W1tp && dx, MDVIT_E_SHAPE, "mlp_rc_dgrad: null operand");
    MDVIT_CHECK_ARG(aligned16(gm) && aligned16(x) && aligned16(W1p) && aligned16(b1) && aligned16(W2tp) && aligned16(W1tp) && aligned16(dx), MDVIT_E_ALIGN,
                    "mlp_rc_dgrad: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp_rc_dgrad: dropout index space exceeds 2^32");
    RcArgs a;
    memset(&a, 0, sizeof(a));
    a.gm = gm; a.x = x; a.W1p = (const uint16_t*)W1p; a.b1 = b1; a.W2tp = (const uint16_t*)W2tp; a.W1tp = (const uint16_t*)W1tp; a.dx = dx;
    rc_fill(a, M, Hd, drop_p, key1_0, key1_1, 0, 0, drop_seed);
    constexpr int NW = 4;
    const int smem = 3 * 4 * (32 * 128) + 3 * 2 * (64 * 64) + Hd * 4;
    static bool flags[64] = {false};
    static bool flags2[64] = {false};
    int rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_dgrad_kernel<64, NW, false>), 3 * 4 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, flags);
    if (rc == MDVIT_OK) rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_dgrad_kernel<64, NW, true>), 3 * 4 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, flags2);
    if (rc != MDVIT_OK) return rc;
    if (g_rc_planes == 1) {
        constexpr int CAP = 3 * 4 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4;
        rc = a.drop ? rc_launch<&mlp_rc_dgrad_kernel<64, NW, true, true>>(CAP, dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, (hipStream_t)stream, a)
                    : rc_launch<&mlp_rc_dgrad_kernel<64, NW, false, true>>(CAP, dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, (hipStream_t)stream, a);
        if (rc != MDVIT_OK) return rc;
    } else if (a.drop) hipLaunchKernelGGL((mlp_rc_dgrad_kernel<64, NW, true>), dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_rc_dgrad_kernel<64, NW, false>), dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

static int rc_wgrad_groups(int M) {
    const int ntiles = (M + 31) / 32;
    int g = (ntiles + 3) / 4;                      // >= 4 tiles per group where the problem allows it
    g = ((g + 7) / 8) * 8;
    return g < 8 ? 8 : (g > 128 ? 128 : g);
}

/* mdvit_linear_rc with the LayerNorm in front of it fused into the prologue (LN1 -> qkv of SerialBlock_adapt, mdvit.py:286-288,352): x is the
 * LayerNorm's input [M, K] (contiguous), gamma / beta [groups, K] (group = row / (M / groups)); the kernel writes mean / rstd [M] and the
 * normalised rows ln_out [M, K] (the qkv weight-gradient GEMM's operand) and multiplies them -- mdvit_layernorm_fwd's arithmetic, sum for sum. */
extern "C" int mdvit_linear_rc_ln(const float* x, const float* gamma, const float* beta, int32_t groups, float eps, float* mean, float* rstd, float* ln_out,
                                  const void* Wp, int64_t wplane, const float* bias, float* y, int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream) {
    MDVIT_CHECK_ARG(K == 64 || K == 128, MDVIT_E_SHAPE, "linear_rc_ln: built for K = 64 / 128 (got %d)", K);
    MDVIT_CHECK_ARG(M > 0 && N >= 32 && N % 32 == 0 && N <= 4096 && groups > 0 && M % groups == 0, MDVIT_E_SHAPE, "linear_rc_ln: bad shape M=%d N=%d groups=%d", M, N, groups);
    MDVIT_CHECK_ARG(x && gamma && beta && mean && rstd && ln_out && Wp && y && ldc >= N && ldc % 4 == 0, MDVIT_E_SHAPE, "linear_rc_ln: bad operands");
    MDVIT_CHECK_ARG(aligned16(x) && aligned16(gamma) && aligned16(beta) && aligned16(ln_out) && aligned16(Wp) && aligned16(y) && aligned16(bias), MDVIT_E_ALIGN,
                    "linear_rc_ln: operands must be 16-byte aligned");
    LinArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.lda = K; a.Wp = (const uint16_t*)Wp; a.wplane = wplane; a.bias = bias; a.y = y; a.ldc = ldc; a.M = M; a.N = N; a.rows_per_scale = 1; a.inv_keep = 1.f;
    a.ln_g = gamma; a.ln_b = beta; a.ln_mean = mean; a.ln_rstd = rstd; a.ln_out = ln_out; a.ln_eps = eps; a.ln_rows_per_group = M / groups;
    const int smem = 3 * 2 * (32 * K * 2) + N * 4;
    hipStream_t s = (hipStream_t)stream;
    const int nw = cdiv(M, 128) >= 512 ? 4 : 2;
    const dim3 grid(cdiv(M, nw * 32)), block(nw * 64);
    if (K == 64) {
        if (nw == 4) hipLaunchKernelGGL((lin_rc_kernel<64, 4, false, false, true>), grid, block, smem, s, a);
        else hipLaunchKernelGGL((lin_rc_kernel<64, 2, false, false, true>), grid, block, smem, s, a);
    } else {
        if (nw == 4) hipLaunchKernelGGL((lin_rc_kernel<128, 4, false, false, true>), grid, block, smem, s, a);
        else hipLaunchKernelGGL((lin_rc_kernel<128, 2, false, false, true>), grid, block, smem, s, a);
    }
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

/* The MLP forward on 16-token waves, built for C = 64 and C = 128 (mpvit.py:71-78 inside mdvit.py:357-360); h != NULL also writes
 * h = drop1(gelu(x W1^T + b1)) [M, hidden] for the fc2 weight-gradient GEMM (the forward never re-reads it). */
static int rc16_fwd_impl(int hbf, const float* x, const void* W1p, const float* b1, const void* W2p, const float* b2, const float* res, const float* rowscale,
                                  int32_t rows_per_scale, float* h, float* y, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1,
                                  uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(C == 64 || C == 128, MDVIT_E_SHAPE, "mlp_rc16_fwd: built for C = 64 / 128 (got %d)", C);
    MDVIT_CHECK_ARG(M > 0 && Hd >= 64 && Hd % 32 == 0 && Hd <= 4096, MDVIT_E_SHAPE, "mlp_rc16_fwd: need M > 0, hidden %% 32 == 0, hidden <= 4096 (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(x && W1p && b1 && W2p && b2 && res && y, MDVIT_E_SHAPE, "mlp_rc16_fwd: null operand");
    MDVIT_CHECK_ARG(aligned16(x) && aligned16(W1p) && aligned16(b1) && aligned16(W2p) && aligned16(b2) && aligned16(res) && aligned16(y) && aligned16(h), MDVIT_E_ALIGN,
                    "mlp_rc16_fwd: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp_rc16_fwd: dropout index space exceeds 2^32");
    RcArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.W1p = (const uint16_t*)W1p; a.b1 = b1; a.W2p = (const uint16_t*)W2p; a.b2 = b2; a.res = res; a.rowscale = rowscale; a.y = y; a.h = h;
    a.hbf = hbf && h;
    a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    rc_fill(a, M, Hd, drop_p, key1_0, key1_1, key2_0, key2_1, drop_seed);
    constexpr int NW = 8;
    const int smem = 3 * 2 * (32 * C * 2) + 3 * 2 * (C * 64) + Hd * 4;
    const dim3 grid(cdiv(M, NW * 16)), block(NW * 64);
    hipStream_t s = (hipStream_t)stream;
#define RC16_FWD_LAUNCH(CV, OCCV, DROPV, STOREV)                                                                                     \
    do {                                                                                                                             \
        static bool fl[64] = {false};                                                                                                \
        const int rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc16_fwd_kernel<CV, NW, OCCV, DROPV, STOREV>),                  \
                                  3 * 2 * (32 * CV * 2) + 3 * 2 * (CV * 64) + 4096 * 4, fl);                                         \
        if (rc != MDVIT_OK) return rc;                                                                                               \
        if (g_rc_planes == 1) {                                                                                                      \
            const int rc1 = rc_launch<&mlp_rc16_fwd_kernel<CV, NW, OCCV, DROPV, STOREV, false, true>>(3 * 2 * (32 * CV * 2) + 3 * 2 * (CV * 64) + 4096 * 4, grid, block, smem, s, a); \
            if (rc1 != MDVIT_OK) return rc1;                                                                                         \
        } else hipLaunchKernelGGL((mlp_rc16_fwd_kernel<CV, NW, OCCV, DROPV, STOREV>), grid, block, smem, s, a);                     \
    } while (0)
    if (C == 128 && g_rc_fwd128_variant == 32 && !a.hbf) {
        // 32-token waves on 32x32x16 tiles: the forward's registers allow it (x fragments 64 + y accumulator 64), and every weight fragment read
        // from LDS then serves twice the tokens of the 16-token form
        const int smem32 = 3 * 2 * (32 * 256) + 3 * 2 * (128 * 64) + Hd * 4;
        const dim3 grid32(cdiv(M, 8 * 32)), block32(512);
#define RC32_FWD_LAUNCH(DROPV, STOREV)                                                                                               \
    do {                                                                                                                             \
        static bool fl[64] = {false};                                                                                                \
        const int rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_fwd3_kernel<128, 8, DROPV, 2, STOREV>),                      \
                                  3 * 2 * (32 * 256) + 3 * 2 * (128 * 64) + 4096 * 4, fl);                                           \
        if (rc != MDVIT_OK) return rc;                                                                                               \
        hipLaunchKernelGGL((mlp_rc_fwd3_kernel<128, 8, DROPV, 2, STOREV>), grid32, block32, smem32, s, a);                           \
    } while (0)
        if (a.drop) { if (h) RC32_FWD_LAUNCH(true, true); else RC32_FWD_LAUNCH(true, false); }
        else { if (h) RC32_FWD_LAUNCH(false, true); else RC32_FWD_LAUNCH(false, false); }
#undef RC32_FWD_LAUNCH
    } else if (C == 128) {
        if (a.drop) { if (h) RC16_FWD_LAUNCH(128, 2, true, true); else RC16_FWD_LAUNCH(128, 2, true, false); }
        else { if (h) RC16_FWD_LAUNCH(128, 2, false, true); else RC16_FWD_LAUNCH(128, 2, false, false); }
    } else {
        if (a.drop) { if (h) RC16_FWD_LAUNCH(64, 4, true, true); else RC16_FWD_LAUNCH(64, 4, true, false); }
        else { if (h) RC16_FWD_LAUNCH(64, 4, false, true); else RC16_FWD_LAUNCH(64, 4, false, false); }
    }
#undef RC16_FWD_LAUNCH
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
extern "C" int mdvit_mlp_rc16_fwd(const float* x, const void* W1p, const float* b1, const void* W2p, const float* b2, const float* res, const float* rowscale,
                                  int32_t rows_per_scale, float* h, float* y, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1,
                                  uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed, void* stream) {
    return rc16_fwd_impl(0, x, W1p, b1, W2p, b2, res, rowscale, rows_per_scale, h, y, M, C, Hd, drop_p, key1_0, key1_1, key2_0, key2_1, drop_seed, stream);
}
/* the same with h stored as bf16 ([M, hidden] 2-byte elements behind the float pointer: the hi plane the kernel forms anyway) */
extern "C" int mdvit_mlp_rc16_fwd_hbf16(const float* x, const void* W1p, const float* b1, const void* W2p, const float* b2, const float* res, const float* rowscale,
                                  int32_t rows_per_scale, float* h, float* y, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1,
                                  uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed, void* stream) {
    return rc16_fwd_impl(1, x, W1p, b1, W2p, b2, res, rowscale, rows_per_scale, h, y, M, C, Hd, drop_p, key1_0, key1_1, key2_0, key2_1, drop_seed, stream);
}

/* The MLP forward with the LayerNorm in front of it fused into the prologue (LN2 -> Mlp of SerialBlock_adapt, mdvit.py:356-360): x2 [M, C] is the LayerNorm's
 * INPUT and the residual; writes mean / rstd [M], the normalised rows ln_out [M, C] (operand of the backward kernels) and y.  C = 64: mlp_rc_fwd3 (no
 * [tokens, hidden] tensor); C = 128: the 16-token kernel, h != NULL written as in mdvit_mlp_rc16_fwd.  mdvit_layernorm_fwd's arithmetic, sum for sum. */
static int rc_fwd_ln_impl(int hbf, const float* x2, const float* gamma, const float* beta, int32_t groups, float eps, float* mean, float* rstd, float* ln_out,
                                   const void* W1p, const float* b1, const void* W2p, const float* b2, const float* rowscale, int32_t rows_per_scale, float* h,
                                   float* y, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, uint32_t key2_0, uint32_t key2_1,
                                   const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(C == 64 || C == 128, MDVIT_E_SHAPE, "mlp_rc_fwd_ln: built for C = 64 / 128 (got %d)", C);
    MDVIT_CHECK_ARG(M > 0 && Hd >= 64 && Hd % 64 == 0 && Hd <= 4096 && groups > 0 && M % groups == 0, MDVIT_E_SHAPE, "mlp_rc_fwd_ln: bad shape M=%d hidden=%d groups=%d", M, Hd, groups);
    MDVIT_CHECK_ARG(x2 && gamma && beta && mean && rstd && ln_out && W1p && b1 && W2p && b2 && y && (C == 128 || h == nullptr), MDVIT_E_SHAPE, "mlp_rc_fwd_ln: bad operands");
    MDVIT_CHECK_ARG(aligned16(x2) && aligned16(gamma) && aligned16(beta) && aligned16(ln_out) && aligned16(W1p) && aligned16(b1) && aligned16(W2p) && aligned16(b2) &&
                        aligned16(y) && aligned16(h), MDVIT_E_ALIGN, "mlp_rc_fwd_ln: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp_rc_fwd_ln: dropout index space exceeds 2^32");
    RcArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x2; a.res = x2; a.W1p = (const uint16_t*)W1p; a.b1 = b1; a.W2p = (const uint16_t*)W2p; a.b2 = b2; a.rowscale = rowscale; a.y = y; a.h = h;
    a.hbf = hbf && h;
    a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1;
    a.ln = LnPro{gamma, beta, mean, rstd, ln_out, eps, M / groups};
    rc_fill(a, M, Hd, drop_p, key1_0, key1_1, key2_0, key2_1, drop_seed);
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) {
        constexpr int NW = 4;
        const int smem = 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + Hd * 4;
        static bool f0[64] = {false}, f1[64] = {false};
        int rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_fwd3_kernel<64, NW, false, 3, false, true>), 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, f0);
        if (rc == MDVIT_OK) rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_fwd3_kernel<64, NW, true, 3, false, true>), 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, f1);
        if (rc != MDVIT_OK) return rc;
        if (g_rc_planes == 1) {
            constexpr int CAP = 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4;
            rc = a.drop ? rc_launch<&mlp_rc_fwd3_kernel<64, NW, true, 3, false, true, true>>(CAP, dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, s, a)
                        : rc_launch<&mlp_rc_fwd3_kernel<64, NW, false, 3, false, true, true>>(CAP, dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, s, a);
            if (rc != MDVIT_OK) return rc;
        } else if (a.drop) hipLaunchKernelGGL((mlp_rc_fwd3_kernel<64, NW, true, 3, false, true>), dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, s, a);
        else hipLaunchKernelGGL((mlp_rc_fwd3_kernel<64, NW, false, 3, false, true>), dim3(cdiv(M, NW * 32)), dim3(NW * 64), smem, s, a);
    } else {
        constexpr int NW = 8;
        const int smem = 3 * 2 * (32 * 128 * 2) + 3 * 2 * (128 * 64) + Hd * 4;
        const dim3 grid(cdiv(M, NW * 16)), block(NW * 64);
#define RC16_FWD_LN_LAUNCH(DROPV, STOREV)                                                                                            \
    do {                                                                                                                             \
        static bool fl[64] = {false};                                                                                                \
        const int rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc16_fwd_kernel<128, NW, 2, DROPV, STOREV, true>),              \
                                  3 * 2 * (32 * 128 * 2) + 3 * 2 * (128 * 64) + 4096 * 4, fl);                                       \
        if (rc != MDVIT_OK) return rc;                                                                                               \
        if (g_rc_planes == 1) {                                                                                                      \
            const int rc1 = rc_launch<&mlp_rc16_fwd_kernel<128, NW, 2, DROPV, STOREV, true, true>>(3 * 2 * (32 * 256) + 3 * 2 * (128 * 64) + 4096 * 4, grid, block, smem, s, a); \
            if (rc1 != MDVIT_OK) return rc1;                                                                                         \
        } else                                                                                                                       \
        hipLaunchKernelGGL((mlp_rc16_fwd_kernel<128, NW, 2, DROPV, STOREV, true>), grid, block, smem, s, a);                         \
    } while (0)
        if (a.drop) { if (h) RC16_FWD_LN_LAUNCH(true, true); else RC16_FWD_LN_LAUNCH(true, false); }
        else { if (h) RC16_FWD_LN_LAUNCH(false, true); else RC16_FWD_LN_LAUNCH(false, false); }
#undef RC16_FWD_LN_LAUNCH
    }
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
extern "C" int mdvit_mlp_rc_fwd_ln(const float* x2, const float* gamma, const float* beta, int32_t groups, float eps, float* mean, float* rstd, float* ln_out,
                                   const void* W1p, const float* b1, const void* W2p, const float* b2, const float* rowscale, int32_t rows_per_scale, float* h,
                                   float* y, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, uint32_t key2_0, uint32_t key2_1,
                                   const uint32_t* drop_seed, void* stream) {
    return rc_fwd_ln_impl(0, x2, gamma, beta, groups, eps, mean, rstd, ln_out, W1p, b1, W2p, b2, rowscale, rows_per_scale, h, y, M, C, Hd, drop_p, key1_0, key1_1, key2_0, key2_1, drop_seed, stream);
}
/* the same with h (C = 128) stored as bf16 */
extern "C" int mdvit_mlp_rc_fwd_ln_hbf16(const float* x2, const float* gamma, const float* beta, int32_t groups, float eps, float* mean, float* rstd, float* ln_out,
                                   const void* W1p, const float* b1, const void* W2p, const float* b2, const float* rowscale, int32_t rows_per_scale, float* h,
                                   float* y, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, uint32_t key2_0, uint32_t key2_1,
                                   const uint32_t* drop_seed, void* stream) {
    return rc_fwd_ln_impl(1, x2, gamma, beta, groups, eps, mean, rstd, ln_out, W1p, b1, W2p, b2, rowscale, rows_per_scale, h, y, M, C, Hd, drop_p, key1_0, key1_1, key2_0, key2_1, drop_seed, stream);
}

/* The C = 128 (and C = 64) MLP's backward data path on 16-token waves: dx = ((gm W2) * gelu'(x W1^T + b1) * mask1) W1 in one kernel; du != NULL
 * also writes the hidden-layer gradient [M, hidden] (the operand of the two weight-gradient GEMMs of the full sweep). */
static int rc16_dgrad_impl(int hbf, const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* du, float* dx,
                                    int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(C == 64 || C == 128, MDVIT_E_SHAPE, "mlp_rc16_dgrad: built for C = 64 / 128 (got %d)", C);
    MDVIT_CHECK_ARG(M > 0 && Hd >= 64 && Hd % 32 == 0 && Hd <= 4096, MDVIT_E_SHAPE, "mlp_rc16_dgrad: need M > 0, hidden %% 32 == 0, hidden <= 4096 (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(gm && x && W1p && b1 && W2tp && W1tp && dx, MDVIT_E_SHAPE, "mlp_rc16_dgrad: null operand");
    MDVIT_CHECK_ARG(aligned16(gm) && aligned16(x) && aligned16(W1p) && aligned16(b1) && aligned16(W2tp) && aligned16(W1tp) && aligned16(dx) && aligned16(du), MDVIT_E_ALIGN,
                    "mlp_rc16_dgrad: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp_rc16_dgrad: dropout index space exceeds 2^32");
    RcArgs a;
    memset(&a, 0, sizeof(a));
    a.gm = gm; a.x = x; a.W1p = (const uint16_t*)W1p; a.b1 = b1; a.W2tp = (const uint16_t*)W2tp; a.W1tp = (const uint16_t*)W1tp; a.dx = dx; a.du = du;
    a.hbf = hbf && du;
    rc_fill(a, M, Hd, drop_p, key1_0, key1_1, 0, 0, drop_seed);
    constexpr int NW = 8;
    const int wbytes = 3 * 4 * (32 * C * 2) + 3 * 2 * (C * 64);
    const int smem = wbytes + Hd * 4;
    const dim3 grid(cdiv(M, NW * 16)), block(NW * 64);
    hipStream_t s = (hipStream_t)stream;
#define RC16_DGRAD_LAUNCH(CV, OCCV, DROPV, STOREV)                                                                                   \
    do {                                                                                                                             \
        static bool fl[64] = {false};                                                                                                \
        const int rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc16_dgrad_kernel<CV, NW, OCCV, DROPV, STOREV>),                \
                                  3 * 4 * (32 * CV * 2) + 3 * 2 * (CV * 64) + 4096 * 4, fl);                                         \
        if (rc != MDVIT_OK) return rc;                                                                                               \
        if (g_rc_planes == 1) {                                                                                                      \
            const int rc1 = rc_launch<&mlp_rc16_dgrad_kernel<CV, NW, OCCV, DROPV, STOREV, true>>(3 * 4 * (32 * CV * 2) + 3 * 2 * (CV * 64) + 4096 * 4, grid, block, smem, s, a); \
            if (rc1 != MDVIT_OK) return rc1;                                                                                         \
        } else                                                                                                                       \
        hipLaunchKernelGGL((mlp_rc16_dgrad_kernel<CV, NW, OCCV, DROPV, STOREV>), grid, block, smem, s, a);                           \
    } while (0)
    if (C == 128) {
        if (a.drop) { if (du) RC16_DGRAD_LAUNCH(128, 2, true, true); else RC16_DGRAD_LAUNCH(128, 2, true, false); }
        else { if (du) RC16_DGRAD_LAUNCH(128, 2, false, true); else RC16_DGRAD_LAUNCH(128, 2, false, false); }
    } else {
        if (a.drop) { if (du) RC16_DGRAD_LAUNCH(64, 4, true, true); else RC16_DGRAD_LAUNCH(64, 4, true, false); }
        else { if (du) RC16_DGRAD_LAUNCH(64, 4, false, true); else RC16_DGRAD_LAUNCH(64, 4, false, false); }
    }
#undef RC16_DGRAD_LAUNCH
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
extern "C" int mdvit_mlp_rc16_dgrad(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* du, float* dx,
                                    int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, void* stream) {
    return rc16_dgrad_impl(0, gm, x, W1p, b1, W2tp, W1tp, du, dx, M, C, Hd, drop_p, key1_0, key1_1, drop_seed, stream);
}
/* the same with du stored as bf16 */
extern "C" int mdvit_mlp_rc16_dgrad_hbf16(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* du, float* dx,
                                    int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, void* stream) {
    return rc16_dgrad_impl(1, gm, x, W1p, b1, W2tp, W1tp, du, dx, M, C, Hd, drop_p, key1_0, key1_1, drop_seed, stream);
}

/* y[M, N] = x[M, K] Wp^T (+ bias) for K = 64 / 128, N % 32 == 0; with `residual` also x dropout(key) x rowscale[row / rows_per_scale] + residual
 * (the GEMM's FULL epilogue, same mask indices row * N + col).  Wp: the bf16 hi / lo planes [2][N][K] of the weight (or of its transpose, for a
 * data gradient), plane stride wplane elements. */
extern "C" int mdvit_linear_rc(const float* x, int64_t lda, const void* Wp, int64_t wplane, const float* bias, float* y, int64_t ldc, int32_t M, int32_t N, int32_t K,
                               float drop_p, uint32_t key0, uint32_t key1, const float* rowscale, int32_t rows_per_scale, const float* residual, int64_t ldr,
                               const uint32_t* drop_seed, void* stream) {
    MDVIT_CHECK_ARG(K == 64 || K == 128, MDVIT_E_SHAPE, "linear_rc: built for K = 64 / 128 (got %d)", K);
    MDVIT_CHECK_ARG(M > 0 && N >= 32 && N % 32 == 0 && N <= 4096, MDVIT_E_SHAPE, "linear_rc: need M > 0, N %% 32 == 0, N <= 4096 (M=%d N=%d)", M, N);
    MDVIT_CHECK_ARG(x && Wp && y && lda >= K && ldc >= N && lda % 4 == 0 && ldc % 4 == 0 && (!residual || (ldr >= N && ldr % 4 == 0)), MDVIT_E_SHAPE, "linear_rc: bad operands");
    MDVIT_CHECK_ARG(aligned16(x) && aligned16(Wp) && aligned16(y) && aligned16(bias) && aligned16(residual), MDVIT_E_ALIGN, "linear_rc: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (!(drop_p > 0.f) || (long)M * N < (1L << 32)), MDVIT_E_SHAPE, "linear_rc: bad dropout arguments");
    MDVIT_CHECK_ARG(residual || (drop_p == 0.f && rowscale == nullptr), MDVIT_E_SHAPE, "linear_rc: dropout / row scale come with the residual epilogue");
    LinArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.lda = lda; a.Wp = (const uint16_t*)Wp; a.wplane = wplane; a.bias = bias; a.y = y; a.ldc = ldc; a.residual = residual; a.ldr = ldr;
    a.rowscale = rowscale; a.rows_per_scale = rows_per_scale > 0 ? rows_per_scale : 1; a.M = M; a.N = N;
    a.k0 = key0; a.k1 = key1; a.thresh = mdvit_drop_thresh(drop_p); a.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f; a.seed = drop_seed;
    const int smem = 3 * 2 * (32 * K * 2) + N * 4;
    hipStream_t s = (hipStream_t)stream;
    const bool full = residual != nullptr, drop = drop_p > 0.f;
    // 128 tokens per workgroup; 64 while that leaves the chip short of two workgroups per CU (each workgroup streams the whole weight once)
    const int nw = cdiv(M, 128) >= 512 ? 4 : 2;
    const dim3 grid(cdiv(M, nw * 32)), block(nw * 64);
#define LIN_RC_LAUNCH(KV, FULLV, DROPV)                                                                          \
    do {                                                                                                         \
        if (nw == 4) hipLaunchKernelGGL((lin_rc_kernel<KV, 4, FULLV, DROPV>), grid, block, smem, s, a);          \
        else hipLaunchKernelGGL((lin_rc_kernel<KV, 2, FULLV, DROPV>), grid, block, smem, s, a);                  \
    } while (0)
    if (K == 64) {
        if (full) { if (drop) LIN_RC_LAUNCH(64, true, true); else LIN_RC_LAUNCH(64, true, false); }
        else LIN_RC_LAUNCH(64, false, false);
    } else {
        if (full) { if (drop) LIN_RC_LAUNCH(128, true, true); else LIN_RC_LAUNCH(128, true, false); }
        else LIN_RC_LAUNCH(128, false, false);
    }
#undef LIN_RC_LAUNCH
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" size_t mdvit_mlp_rc_wgrad_ws_bytes(int32_t M, int32_t C, int32_t Hd) {
    return sizeof(float) * (size_t)rc_wgrad_groups(M) * (2u * (size_t)Hd * C + Hd);
}

extern "C" int mdvit_mlp_rc_wgrad(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, float* dW1, float* db1, float* dW2,
                                  void* ws, size_t ws_bytes, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1,
                                  const uint32_t* drop_seed, int32_t accumulate, void* stream) {
    MDVIT_CHECK_ARG(C == 64, MDVIT_E_SHAPE, "mlp_rc_wgrad: built for C = 64 (got %d)", C);
    MDVIT_CHECK_ARG(M > 0 && Hd >= 256 && Hd % 256 == 0, MDVIT_E_SHAPE, "mlp_rc_wgrad: need M > 0, hidden %% 256 == 0 (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(gm && x && W1p && b1 && W2tp && dW1 && db1 && dW2, MDVIT_E_SHAPE, "mlp_rc_wgrad: null operand");
    MDVIT_CHECK_ARG(aligned16(gm) && aligned16(x) && aligned16(W1p) && aligned16(W2tp) && aligned16(dW1) && aligned16(db1) && aligned16(dW2) && aligned16(ws),
                    MDVIT_E_ALIGN, "mlp_rc_wgrad: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp_rc_wgrad: dropout index space exceeds 2^32");
    const size_t need = mdvit_mlp_rc_wgrad_ws_bytes(M, C, Hd);
    MDVIT_CHECK_ARG(ws && ws_bytes >= need, MDVIT_E_WORKSPACE, "mlp_rc_wgrad: workspace too small: need %zu bytes (mdvit_mlp_rc_wgrad_ws_bytes), got %zu", need, ws_bytes);
    RcArgs a;
    memset(&a, 0, sizeof(a));
    a.gm = gm; a.x = x; a.W1p = (const uint16_t*)W1p; a.b1 = b1; a.W2tp = (const uint16_t*)W2tp; a.part = (float*)ws;
    rc_fill(a, M, Hd, drop_p, key1_0, key1_1, 0, 0, drop_seed);
    a.groups = rc_wgrad_groups(M);
    a.tiles_per_group = cdiv((M + 31) / 32, a.groups);
    const int roles = Hd / 256;
    if (g_rc_planes == 1) {
        if (a.drop) hipLaunchKernelGGL((mlp_rc_wgrad_kernel<64, true, false, true>), dim3(a.groups * roles), dim3(512), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((mlp_rc_wgrad_kernel<64, false, false, true>), dim3(a.groups * roles), dim3(512), 0, (hipStream_t)stream, a);
    } else if (a.drop) hipLaunchKernelGGL((mlp_rc_wgrad_kernel<64, true>), dim3(a.groups * roles), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_rc_wgrad_kernel<64, false>), dim3(a.groups * roles), dim3(512), 0, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    const int n0 = Hd * C, n2 = Hd;
    hipLaunchKernelGGL(rc_reduce_kernel, dim3(cdiv((2L * n0 + n2) / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, a.groups, (long)(2L * n0 + n2), n0, dW1,
                       n0, dW2, n2, db1, accumulate);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

/* The whole backward of the C = 64 MLP in ONE kernel + its partial-sum fold (round 5): mdvit_mlp_rc_dgrad and mdvit_mlp_rc_wgrad recompute u = x W1^T + b1, d = gm W2 and
 * the activation twice; here they are formed once and feed all three products -- dW1 = du^T x, dW2 = gm^T h (as mdvit_mlp_rc_wgrad, bit for bit) and dx = du W1.
 * dx_parts [hidden / 256][M][C]: one partial of dx per 256-wide hidden role (the consumer adds them: mdvit_layernorm_bwd's dy2, mdvit_sum_batch); hidden in {256, 512}. */
extern "C" int mdvit_mlp_rc_bwd(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* dx_parts,
                                float* dW1, float* db1, float* dW2, void* ws, size_t ws_bytes, int32_t M, int32_t C, int32_t Hd, float drop_p,
                                uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, int32_t accumulate, void* stream) {
    MDVIT_CHECK_ARG(C == 64, MDVIT_E_SHAPE, "mlp_rc_bwd: built for C = 64 (got %d)", C);
    MDVIT_CHECK_ARG(M > 0 && (Hd == 256 || Hd == 512), MDVIT_E_SHAPE, "mlp_rc_bwd: need M > 0, hidden in {256, 512} (M=%d hidden=%d)", M, Hd);
    MDVIT_CHECK_ARG(gm && x && W1p && b1 && W2tp && W1tp && dx_parts && dW1 && db1 && dW2, MDVIT_E_SHAPE, "mlp_rc_bwd: null operand");
    MDVIT_CHECK_ARG(aligned16(gm) && aligned16(x) && aligned16(W1p) && aligned16(W2tp) && aligned16(W1tp) && aligned16(dx_parts) && aligned16(dW1) && aligned16(db1) &&
                    aligned16(dW2) && aligned16(ws), MDVIT_E_ALIGN, "mlp_rc_bwd: operands must be 16-byte aligned");
    MDVIT_CHECK_ARG(!(drop_p > 0.f) || (long)M * Hd < (1L << 32), MDVIT_E_SHAPE, "mlp_rc_bwd: dropout index space exceeds 2^32");
    const size_t need = mdvit_mlp_rc_wgrad_ws_bytes(M, C, Hd);
    MDVIT_CHECK_ARG(ws && ws_bytes >= need, MDVIT_E_WORKSPACE, "mlp_rc_bwd: workspace too small: need %zu bytes (mdvit_mlp_rc_wgrad_ws_bytes), got %zu", need, ws_bytes);
    RcArgs a;
    memset(&a, 0, sizeof(a));
    a.gm = gm; a.x = x; a.W1p = (const uint16_t*)W1p; a.b1 = b1; a.W2tp = (const uint16_t*)W2tp; a.W1tp = (const uint16_t*)W1tp; a.dx = dx_parts; a.part = (float*)ws;
    rc_fill(a, M, Hd, drop_p, key1_0, key1_1, 0, 0, drop_seed);
    a.groups = rc_wgrad_groups(M);
    a.tiles_per_group = cdiv((M + 31) / 32, a.groups);
    const int roles = Hd / 256;
    static bool f0[64] = {false}, f1[64] = {false};
    int rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_wgrad_kernel<64, true, true>), RC_BWD_LDS, f0);
    if (rc == MDVIT_OK) rc = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc_wgrad_kernel<64, false, true>), RC_BWD_LDS, f1);
    if (rc != MDVIT_OK) return rc;
    if (g_rc_planes == 1) {
        rc = a.drop ? rc_launch<&mlp_rc_wgrad_kernel<64, true, true, true>>(RC_BWD_LDS, dim3(a.groups * roles), dim3(512), RC_BWD_LDS, (hipStream_t)stream, a)
                    : rc_launch<&mlp_rc_wgrad_kernel<64, false, true, true>>(RC_BWD_LDS, dim3(a.groups * roles), dim3(512), RC_BWD_LDS, (hipStream_t)stream, a);
        if (rc != MDVIT_OK) return rc;
    } else if (a.drop) hipLaunchKernelGGL((mlp_rc_wgrad_kernel<64, true, true>), dim3(a.groups * roles), dim3(512), RC_BWD_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((mlp_rc_wgrad_kernel<64, false, true>), dim3(a.groups * roles), dim3(512), RC_BWD_LDS, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    const int n0 = Hd * C, n2 = Hd;
    hipLaunchKernelGGL(rc_reduce_kernel, dim3(cdiv((2L * n0 + n2) / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, a.groups, (long)(2L * n0 + n2), n0, dW1,
                       n0, dW2, n2, db1, accumulate);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

