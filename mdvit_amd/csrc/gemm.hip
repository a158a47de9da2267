// GEMM family on the CDNA4 matrix cores; fp32 in HBM, two arithmetic modes:
//   bf16x3 (default): every operand split hi + lo into bf16 while it is staged into LDS, hi*hi + hi*lo + lo*hi on
//                     v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~4e-6 relative);
//   fp32            : v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain (157 TF peak).
// One kernel template covers the forward linear layers and the data gradients (NT; dgrad reads the cached W^T) and the
// weight gradients (TN, split along the token axis into dense slabs + a fixed-order reduce; the bias gradient rides on its
// A stream), with the element-wise neighbours of each GEMM fused into the epilogue:
//   +bias | exact-erf GELU (dual store u, h=dropout(gelu(u))) | dropout | DropPath row scale | +residual |
//   x gelu'(u) x dropout-mask (fc2 dgrad).
// Replaces nn.Linear / 1x1 nn.Conv2d / einsum call sites of the reference:
//   mdvit.py:288 (qkv), :310-311 (proj+drop), mpvit.py:71-78 (Mlp), Decoders.py:196,319-331 (1x1 convs).
#include "common.h"
#include <type_traits>
#ifndef MDVIT_NO_DB
#define MDVIT_NO_DB 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// set by mdvit_gemm_f32_grouped around its call of mdvit_gemm_f32 (same thread): the operand triples of the groups
struct GemmGroups { int n; const float* A[MDVIT_GEMM_MAX_GROUPS]; const float* B[MDVIT_GEMM_MAX_GROUPS]; float* C[MDVIT_GEMM_MAX_GROUPS]; const float* bias[MDVIT_GEMM_MAX_GROUPS]; };
static thread_local GemmGroups g_groups = {0, {}, {}, {}, {}};
const GemmGroups* mdvit_gemm_groups_active() { return g_groups.n > 0 ? &g_groups : nullptr; }

namespace {

constexpr int BK = 32;
constexpr int NTHREADS = 256;

struct GemmArgs {
    const float* A; const float* B; float* C; float* C2;
    long lda, ldb, ldc;
    int M, N, K;
    const float* bias;
    // A prologue
    // epilogue
    int epi;                       // MDVIT_EPI_*
    int e_drop; uint32_t e_k0, e_k1, e_thresh; float e_inv_keep;
    const float* e_rowscale; int e_rows_per_scale;
    const float* residual; long ldr;
    const float* gelu_u; long ldu;
    const float* rc_a; long rc_lda; const float* rc_b; long rc_ldb; const float* rc_bias; int rc_k;    // DGELU_RC: u = rc_a rc_b^T + rc_bias
    int splits; int k_per_split;   // split along K: each split writes a dense [M,N] slab, reduced by a second kernel
    float* slab;
    int accumulate;
    float* colsum;                 // TN only: colsum[m] += sum_k A[k][m] (the bias gradient rides on the wgrad's A stream)
    int vec;                       // output rows can take 16-byte vector accesses
    const uint32_t* seed;          // optional device-side dropout seed {s0, s1}: key0 ^= s0, key1 += s1 (graph replays draw fresh masks)
    int tiles_m, tiles_n;
    int conv_c, conv_h, conv_w, conv_ho, conv_wo, conv_stride, conv_dil, conv_up, conv_phase;     // CONV kernels: A is an NHWC image gathered on the fly
    // grouped launch (mdvit_gemm_f32_grouped): blockIdx.z = group; the same problem on ngroups operand triples (no split, plain epilogue, no bias)
    int ngroups; const float* gA[MDVIT_GEMM_MAX_GROUPS]; const float* gB[MDVIT_GEMM_MAX_GROUPS]; float* gC[MDVIT_GEMM_MAX_GROUPS]; const float* gBias[MDVIT_GEMM_MAX_GROUPS];
};



// Bijective XCD-aware remap (guide T1): consecutive logical tiles share an XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// EPI (compile time, so every instantiation carries only its own epilogue -- the unrolled epilogue of a single
// do-everything kernel was ~20k instructions, far beyond the instruction cache):
//   0 PLAIN  : (+bias) | split slab | accumulate        1 GELU_DUAL    2 DGELU (x dropout mask)
//   3 FULL   : +bias, dropout, DropPath row scale, +residual
//   4 DGELU_RC : DGELU whose pre-activation u is not read from HBM but RECOMPUTED in the kernel as a second product
//                u = rc_a rc_b^T + rc_bias over the same output tile (the MLP's fc1: K = C <= 128, a few MFMAs) -- the same
//                slab / MFMA sequence as the forward GEMM, so u is bit-identical to the one the forward computed.  Saves the
//                forward's store of u and this kernel's read of it ([tokens, hidden] each) on the HBM-bound MLPs.
enum { EPI_PLAIN = 0, EPI_GELU2 = 1, EPI_DGELU = 2, EPI_FULL = 3, EPI_DGELU_RC = 4 };

// BF3 ("bf16x3", NT layout only): every fp32 operand is split as x = hi + lo (two RNE bf16 values, |x - hi - lo| <= 2^-18 |x|)
// while it is staged into LDS, and each product runs as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation -- three 32-cycle MFMAs per K=16 instead of eight 64-cycle fp32 MFMAs (5.3x less matrix-core time),
// ~1e-5 relative accuracy.  LDS holds [row][k] bf16 planes (80-byte rows: conflict-free ds_read_b128 fragments).
__device__ __forceinline__ void split_bf16x3(const float4 x, uint2& hi, uint2& lo) {
    f32x2_t a = {x.x, x.y}, b = {x.z, x.w};
    const bf16x2_t ha = __builtin_convertvector(a, bf16x2_t), hb = __builtin_convertvector(b, bf16x2_t);
    const uint32_t hau = __builtin_bit_cast(uint32_t, ha), hbu = __builtin_bit_cast(uint32_t, hb);
    f32x2_t la = {x.x - __uint_as_float(hau << 16), x.y - __uint_as_float(hau & 0xffff0000u)};
    f32x2_t lb = {x.z - __uint_as_float(hbu << 16), x.w - __uint_as_float(hbu & 0xffff0000u)};
    const bf16x2_t lab = __builtin_convertvector(la, bf16x2_t), lbb = __builtin_convertvector(lb, bf16x2_t);
    hi = make_uint2(hau, hbu);
    lo = make_uint2(__builtin_bit_cast(uint32_t, lab), __builtin_bit_cast(uint32_t, lbb));
}

// (x0, x1) = the same column at k and k+1 -> packed bf16 pairs {hi(x0), hi(x1)} and {lo(x0), lo(x1)}
__device__ __forceinline__ void split_pair_bf16x3(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    f32x2_t a = {x0, x1};
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2_t));
    f32x2_t l = {x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xffff0000u)};
    hi = h;
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(l, bf16x2_t));
}

// The kernel body lives in gemm_body.inc and is textually included into both __global__ functions below (a shared __device__
// function inlined into thin wrappers changed the code generation of the fp32 TN variants -- and their results; the body wants the
// kernel's own parameter).  It expects: BM BN WAVES_M WAVES_N TA TB EPI BF3 CONV as compile-time constants and GemmArgs p.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool TA, bool TB, int EPI, bool BF3>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu((BM == 256 && TA) ? 2 : 3, 8))) void gemm_f32_kernel(GemmArgs p) {
    constexpr bool CONV = false;
#include "gemm_body.inc"
}

// the same NT bf16x3 main loop with the A operand gathered from an NHWC image (3x3 taps): y = conv3x3(x, w) without im2col
template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(3, 8))) void gemm_conv3x3_kernel(GemmArgs p) {
    constexpr bool CONV = true, TA = false, TB = true, BF3 = true;
    constexpr int EPI = EPI_PLAIN;
#include "gemm_body.inc"
}

// C[m][n] = sum_s slab[s][m][n] (+ bias[n]).  R lanes share one output quad: lane r adds slabs r, r+R, ... and the R
// partial sums are folded by a shuffle tree -- a fixed order for a given shape, so the result is deterministic.
template <int R>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                                 float* __restrict__ C, long ldc, int M, int N, int splits, int accumulate, int perm_cin) {
    const int NQ = N >> 2;
    const long total = (long)M * NQ, MN = (long)M * N;
    const int r = threadIdx.x % R;
    const long stride = (long)gridDim.x * blockDim.x / R;
    for (long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) / R; e < total; e += stride) {
        const long m = e / NQ;
        const int n = (int)(e % NQ) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = slab + m * N + n;
        for (int sidx = r; sidx < splits; sidx += R) {
            const float4 v = *reinterpret_cast<const float4*>(src + (long)sidx * MN);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
#pragma unroll
        for (int off = R / 2; off > 0; off >>= 1) {
            acc.x += __shfl_down(acc.x, off, R); acc.y += __shfl_down(acc.y, off, R);
            acc.z += __shfl_down(acc.z, off, R); acc.w += __shfl_down(acc.w, off, R);
        }
        if (r == 0) {
            if (bias) { acc.x += bias[n]; acc.y += bias[n + 1]; acc.z += bias[n + 2]; acc.w += bias[n + 3]; }
            if (perm_cin > 0) {            // conv weight gradient: column n = (tap, ci) of the tap-major product goes to [ci][tap] of the PyTorch [Cout, Cin, 3, 3] row
                const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nn = n + j, tap = nn / perm_cin, ci = nn - tap * perm_cin;
                    float* d1 = C + m * ldc + (long)ci * 9 + tap;
                    *d1 = accumulate ? *d1 + a4[j] : a4[j];
                }
                continue;
            }
            float* dst = C + m * ldc + n;
            if (accumulate) { acc.x += dst[0]; acc.y += dst[1]; acc.z += dst[2]; acc.w += dst[3]; }
            if (((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0)) *reinterpret_cast<float4*>(dst) = acc;
            else { dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z; dst[3] = acc.w; }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmArgs& a, int ta, int tb, int epi, int bf3, hipStream_t s) {
    dim3 grid(a.tiles_m * a.tiles_n, a.splits, a.ngroups > 0 ? a.ngroups : 1), block(NTHREADS);
#define MDVIT_GEMM_LAUNCH(TA_, TB_, EPI_, BF3_) \
    MDVIT_TIMED_LAUNCH((gemm_f32_kernel<BM, BN, WM, WN, TA_, TB_, EPI_, BF3_>), grid, block, 0, s, a)
    if (a.conv_c > 0) {
        if (ta || !tb || !bf3 || epi != EPI_PLAIN) return 1;
        MDVIT_TIMED_LAUNCH((gemm_conv3x3_kernel<BM, BN, WM, WN>), grid, block, 0, s, a);
        return 0;
    }
    if (!ta && tb) {                                   // forward (weights [N,K]); with transposed weights also dgrad
        if (bf3) {
            if (epi == EPI_GELU2) MDVIT_GEMM_LAUNCH(false, true, EPI_GELU2, true);
            else if (epi == EPI_DGELU) MDVIT_GEMM_LAUNCH(false, true, EPI_DGELU, true);
            else if (epi == EPI_DGELU_RC) { if (BM == 64 && BN == 64) MDVIT_GEMM_LAUNCH(false, true, (BM == 64 && BN == 64 ? EPI_DGELU_RC : EPI_DGELU), true); else return 1; }
            else if (epi == EPI_FULL) MDVIT_GEMM_LAUNCH(false, true, EPI_FULL, true);
            else if (epi == EPI_PLAIN) MDVIT_GEMM_LAUNCH(false, true, EPI_PLAIN, true);
            else return 1;
        } else {
            if (epi == EPI_GELU2) MDVIT_GEMM_LAUNCH(false, true, EPI_GELU2, false);
            else if (epi == EPI_DGELU_RC) { if (BM == 64 && BN == 64) MDVIT_GEMM_LAUNCH(false, true, (BM == 64 && BN == 64 ? EPI_DGELU_RC : EPI_GELU2), false); else return 1; }
            else if (epi == EPI_FULL) MDVIT_GEMM_LAUNCH(false, true, EPI_FULL, false);
            else if (epi == EPI_PLAIN) MDVIT_GEMM_LAUNCH(false, true, EPI_PLAIN, false);
            else return 1;
        }
    } else if (bf3) {
        if (ta && !tb && epi == EPI_PLAIN) MDVIT_GEMM_LAUNCH(true, false, EPI_PLAIN, true);     // wgrad, pair-staged operands
        else return 1;
    } else if (!ta && !tb) {                           // dgrad
        if (epi == EPI_DGELU) MDVIT_GEMM_LAUNCH(false, false, EPI_DGELU, false);
        else if (epi == EPI_PLAIN) MDVIT_GEMM_LAUNCH(false, false, EPI_PLAIN, false);
        else if (epi == EPI_FULL) MDVIT_GEMM_LAUNCH(false, false, EPI_FULL, false);
        else return 1;
    } else if (ta && !tb) {                            // wgrad
        if (epi != EPI_PLAIN) return 1;
        MDVIT_GEMM_LAUNCH(true, false, EPI_PLAIN, false);
    } else {
        return 1;
    }
#undef MDVIT_GEMM_LAUNCH
    return 0;
}

struct GemmPlan { int cfg, tiles_m, tiles_n, splits, kps; };

// Tile shape and K-split from a small cost model (CU cycles):
//   cfg 0: 128x128 (2 workgroups / CU)   1: 256x64 (narrow outputs)   2: 64x64 (4 workgroups / CU; few or ragged tiles)
// cost = rounds over the chip x workgroups sharing a CU x padded tile work / tile efficiency, plus -- when the K range
// is split into slabs -- the fixed-order slab reduction (a second, HBM-bound kernel).
int g_force_cfg = -1, g_force_splits = 0;      // tuning hook (mdvit_gemm_force_plan); -1 / 0 = planner decides

// cycles charged for the second launch of a split K range (the reduction kernel + the gap in front of it on the stream).  30000 measured +0.6-0.8 % on the
// bs=4 step (tools/ab_values.sh) but moves which products split, i.e. their summation order -- the two-sample BatchNorm of the DeepLabV3 heads' pooling branch
// amplifies that past its golden bound (test_mdvit_two_sweep_step_vs_golden[...deeplab...]: 2.9e-2 against 1.5e-2), so the default stays
static const double g_split_penalty = [] { const char* e = getenv("MDVIT_SPLIT_PENALTY"); return e ? atof(e) : 12000.0; }();

GemmPlan plan_gemm(const MdvitGemmDesc* d) {
    static const int BMs[3] = {128, 256, 64}, BNs[3] = {128, 64, 64}, OCC[3] = {2, 2, 4};
    static const double EFF[3] = {0.8, 0.8, 1.0};       // measured (tools/gemm_sweep.py): the 64x64 tile at 4 workgroups / CU wins almost everywhere
    static const int SPLITS[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256};
    const bool plain = d->epi == MDVIT_EPI_NONE && !(d->e_drop_p > 0.f) && !d->e_rowscale && !d->residual;
    const bool can_split = d->allow_split && plain && d->K >= 512 && (d->N % 4 == 0);
    GemmPlan best; best.cfg = 0; best.tiles_m = cdiv(d->M, 128); best.tiles_n = cdiv(d->N, 128); best.splits = 1;
    best.kps = cdiv(d->K, BK) * BK;
    // The dense 3x3 convolutions of the bridge at 16 images (implicit GEMM, 4096 pixels x 512 / 1024 channels, K = 4608 / 9216): the model below prices the gathered A
    // operand like a matrix and picked the 64 x 64 tile in one K range for 512 -> 1024 (228 us) -- measured (tools/probe/conv_bridge_plans.py,
    // profiles/r05_conv_bridge_plans.txt): 128 x 128 tiles over K ranges that make ~512 workgroups (two per CU) run 512 -> 1024 in 158-165 us, 512 -> 512 in 92 (106),
    // 1024 -> 512 in 158 (157); at 128 images (>= 1024 tiles) the one-range plans stand.  MDVIT_CONV_SPLIT_PLAN=0: the model's choice (A/B).
    static const bool conv_rule = [] { const char* e = getenv("MDVIT_CONV_SPLIT_PLAN"); return !(e && e[0] == '0'); }();
    if (conv_rule && g_force_cfg < 0 && g_force_splits == 0 && d->conv_c > 0 && d->conv_up <= 1 && d->conv_stride == 1 && !d->trans_a && can_split && d->K >= 2304 && !d->rc_a) {
        const long tiles = (long)cdiv(d->M, 128) * cdiv(d->N, 128);
        if (tiles >= 64 && tiles <= 256) {
            const int want = (int)max(1L, min(4L, (512 + tiles / 2) / tiles));
            best.kps = cdiv(cdiv(d->K, want), BK) * BK;
            best.splits = cdiv(d->K, best.kps);
            return best;
        }
    }
    double best_cost = 1e300;
    for (int c = 0; c < 3; ++c) {
        if (g_force_cfg >= 0 && c != g_force_cfg && !d->rc_a) continue;
        if (d->rc_a && c != 2) continue;                 // the recomputing epilogue is built on the 64x64 tile (two accumulator sets)
        const long tm = cdiv(d->M, BMs[c]), tn = cdiv(d->N, BNs[c]);
        const long tiles = tm * tn;
        for (int si = 0; si < (int)(sizeof(SPLITS) / sizeof(int)); ++si) {
            const int want = SPLITS[si];
            if (want > 1 && (!can_split || want > d->K / 256)) break;
            if (g_force_splits > 0 && can_split && want != g_force_splits) continue;
            const int kps = cdiv(cdiv(d->K, want), BK) * BK;
            const int splits = cdiv(d->K, kps);
            const long wgs = tiles * splits;
            const long slots = 256L * OCC[c];
            const double rounds = wgs <= slots ? 1.0 : (double)wgs / (double)slots;   // workgroups do not run in lockstep: no ceil
            // one workgroup alone on a CU: 2*BM*BN*kps flop at ~180 flop/clk; plus a fixed prologue/epilogue cost per workgroup
            // bf16x3: 3 x 32-cycle MFMAs per 32x32x16 on 4 SIMDs, plus the hi/lo split of every staged element (VALU)
            // epilogue: the 64x64 tile prefetches gelu_u / the residual ahead of the K loop; the larger tiles load them in the
            // epilogue, exposed to HBM latency at 2 workgroups per CU
            const bool loads_epi = (d->epi == MDVIT_EPI_DGELU && !d->rc_a) || d->residual != nullptr;
            const double epi_cycles = (loads_epi && c != 2 ? 30.0 : 6.0) * BMs[c] * BNs[c] / 64.0;
            const double wg_cycles = d->precision == 1
                ? 0.0015 * BMs[c] * BNs[c] * (double)kps + (d->trans_a ? 0.12 : 0.06) * (BMs[c] + BNs[c]) * (double)kps + 800.0 + epi_cycles
                : 2.0 * BMs[c] * BNs[c] * (double)kps / (180.0 * EFF[c]) + 800.0 + epi_cycles;
            double cost = rounds * OCC[c] * wg_cycles;
            if (splits > 1) cost += g_split_penalty + (double)(splits + 1) * d->M * d->N * 8.0 / 1250.0;
            if (cost < best_cost) {
                best_cost = cost;
                best.cfg = c; best.tiles_m = (int)tm; best.tiles_n = (int)tn; best.splits = splits; best.kps = kps;
            }
        }
    }
    return best;
}

}  // namespace

// the slab reduction as a library-internal entry (gemm_bp.hip's split-K launches share it)
int mdvit_gemm_splitk_reduce_perm(const float* slab, const float* bias, float* C, long ldc, int M, int N, int splits, int accumulate, int perm_cin, hipStream_t s);
int mdvit_gemm_splitk_reduce(const float* slab, const float* bias, float* C, long ldc, int M, int N, int splits, int accumulate, hipStream_t s) {
    return mdvit_gemm_splitk_reduce_perm(slab, bias, C, ldc, M, N, splits, accumulate, 0, s);
}
// perm_cin > 0: the columns are (tap, ci) pairs of a 3x3 convolution's weight gradient and land at [ci][tap] (the PyTorch layout) -- the relayout launch folded in
int mdvit_gemm_splitk_reduce_perm(const float* slab, const float* bias, float* C, long ldc, int M, int N, int splits, int accumulate, int perm_cin, hipStream_t s) {
    const long total = (long)M * N / 4;
#define MDVIT_REDUCE_LAUNCH(R_) \
    hipLaunchKernelGGL((gemm_splitk_reduce_kernel<R_>), dim3((int)min((total * R_ + 255) / 256, 4096L)), dim3(256), 0, s, \
                       slab, bias, C, ldc, M, N, splits, accumulate, perm_cin)
    if (total >= 65536 || splits < 4) MDVIT_REDUCE_LAUNCH(1);
    else if (total >= 16384 || splits < 16) MDVIT_REDUCE_LAUNCH(4);
    else if (total >= 4096 || splits < 64) MDVIT_REDUCE_LAUNCH(16);
    else MDVIT_REDUCE_LAUNCH(64);
#undef MDVIT_REDUCE_LAUNCH
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

// gemm_tn.hip: the transposing-LDS-read weight-gradient kernel (TN layout, bf16x3 / bf16 arithmetic, plain epilogue)
bool mdvit_gemm_tn_applies(const MdvitGemmDesc* d);
size_t mdvit_gemm_tn_ws_bytes(const MdvitGemmDesc* d);
void mdvit_gemm_tn_plan(const MdvitGemmDesc* d, int* tile_m, int* tile_n, int* splits);
int mdvit_gemm_tn_launch(const MdvitGemmDesc* d, hipStream_t s);

// ---- launch ledger (measurement only, off by default): which kernel every mdvit_gemm_f32 call launched -- the calls of the C-level block entry (block.hip) included,
// which never pass the Python wrapper's event sampler -- with its algorithmic flops / bytes (ops.gemm's formula).  bench.py reads it over ONE untimed step, so that the
// roofline line's launch count and bytes per launch describe the same launch population as the rocprofv3 average it is compared with.
#include <mutex>
namespace {
struct LedgerRow { char name[160]; long launches; double flop, bytes; };
constexpr int LEDGER_ROWS = 512;
LedgerRow g_ledger[LEDGER_ROWS];
int g_ledger_n = 0;
bool g_ledger_on = false, g_ledger_shapes = false;          // enable = 2: one row per (kernel, M, N, K, epilogue operands) instead of one per kernel (tools/gemm_shapes.py)
std::mutex g_ledger_mu;
}
extern "C" int mdvit_gemm_kernel_name(const MdvitGemmDesc* d, char* out, int32_t cap);
extern "C" int mdvit_gemm_ledger(int32_t enable) {
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    if (enable) g_ledger_n = 0;
    g_ledger_on = enable != 0;
    g_ledger_shapes = enable == 2;
    return MDVIT_OK;
}
extern "C" int mdvit_gemm_ledger_read(int32_t index, char* name, int32_t cap, int64_t* launches, double* flop, double* bytes) {
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    MDVIT_CHECK_ARG(name && cap > 0 && launches && flop && bytes, MDVIT_E_SHAPE, "gemm_ledger_read: null argument");
    if (index < 0 || index >= g_ledger_n) return MDVIT_E_SHAPE;            // past the end (no message: the caller's loop condition)
    snprintf(name, cap, "%s", g_ledger[index].name);
    *launches = g_ledger[index].launches; *flop = g_ledger[index].flop; *bytes = g_ledger[index].bytes;
    return MDVIT_OK;
}
static void ledger_note(const MdvitGemmDesc* d) {
    char nm[160];
    if (mdvit_gemm_kernel_name(d, nm, sizeof(nm)) != MDVIT_OK) return;
    if (char* plus = strchr(nm, '+')) *plus = 0;                            // the main kernel (a slab reduction is its own launch)
    std::lock_guard<std::mutex> lk(g_ledger_mu);
    if (g_ledger_shapes) {
        const size_t n = strlen(nm);
        snprintf(nm + n, sizeof(nm) - n, " M=%d N=%d K=%d ta=%d tb=%d%s%s%s%s", d->M, d->N, d->K, (int)d->trans_a, (int)d->trans_b, d->C2 ? " +C2" : "", d->residual ? " +res" : "",
                 d->gelu_u ? " +u" : "", d->rc_a ? " +rc" : "");
    }
    int i = 0;
    while (i < g_ledger_n && strcmp(g_ledger[i].name, nm) != 0) ++i;
    if (i == g_ledger_n) {
        if (g_ledger_n == LEDGER_ROWS) return;
        snprintf(g_ledger[i].name, sizeof(g_ledger[i].name), "%s", nm);
        g_ledger[i].launches = 0; g_ledger[i].flop = 0.0; g_ledger[i].bytes = 0.0;
        ++g_ledger_n;
    }
    const double M = d->M, N = d->N, K = d->K;
    g_ledger[i].launches += 1;
    g_ledger[i].flop += 2.0 * M * N * K;
    g_ledger[i].bytes += 4.0 * (M * K + N * K + M * N * (1 + (d->C2 != nullptr) + (d->residual != nullptr) + (d->gelu_u != nullptr))) + (d->rc_a ? 4.0 * (M + N) * d->rc_k : 0.0);
}

// ---- launch sampler (measurement only, off by default): kernel begin / end timestamps (mdvit_timing_arm's hipExtLaunchKernelGGL events) around a SAMPLE of the launches that
// mdvit_gemm_f32 issues -- from wherever it is called: the Python wrapper never sees the products of mdvit_block_fwd / _bwd, and in round 5 the largest kernel of the traced
// step (the weight-gradient tile, launched from the block entry on the side stream) was invisible to bench.py's roofline for that reason.  symbol NULL / "": every kernel
// symbol (bench.py's scouting step); else that symbol only.  A launch is taken by a hashed 1-in-stride pick over the symbol's launches (the shapes of a kernel cycle with a
// short period: a plain every-stride-th pick locks onto one phase).  mdvit_gemm_sampler_read(i, ...) waits for the sampled launches and returns symbol i's launches seen, launches
// timed and the sum of their durations; MDVIT_E_SHAPE past the last symbol.  Single-threaded use (bench.py runs the sweeps on the calling thread).
namespace {
constexpr int SAMP_MAX = 4096, SAMP_SYMS = 128;
struct SampSym { char name[160]; long seen, timed; double ms; };
SampSym g_samp_sym[SAMP_SYMS];
int g_samp_nsym = 0, g_samp_stride = 0, g_samp_n = 0;
char g_samp_only[160] = {0};
hipEvent_t g_samp_ev[SAMP_MAX][2];
int g_samp_ev_made = 0, g_samp_of[SAMP_MAX];
bool g_samp_folded = true;
}
extern "C" int mdvit_gemm_sampler(const char* symbol, int32_t stride) {
    g_samp_stride = stride > 0 ? stride : 0;
    g_samp_nsym = 0; g_samp_n = 0; g_samp_folded = false;
    snprintf(g_samp_only, sizeof(g_samp_only), "%s", symbol ? symbol : "");
    return MDVIT_OK;
}
extern "C" int mdvit_gemm_sampler_read(int32_t index, char* name, int32_t cap, int64_t* seen, int64_t* timed, double* ms) {
    MDVIT_CHECK_ARG(name && cap > 0 && seen && timed && ms, MDVIT_E_SHAPE, "gemm_sampler_read: null argument");
    if (!g_samp_folded) {
        g_samp_stride = 0;                                   // reading ends the sampling
        for (int i = 0; i < g_samp_n; ++i) {
            float t = 0.f;
            if (hipEventSynchronize(g_samp_ev[i][1]) != hipSuccess || hipEventElapsedTime(&t, g_samp_ev[i][0], g_samp_ev[i][1]) != hipSuccess) { (void)hipGetLastError(); continue; }
            g_samp_sym[g_samp_of[i]].timed += 1; g_samp_sym[g_samp_of[i]].ms += t;
        }
        g_samp_folded = true;
    }
    if (index < 0 || index >= g_samp_nsym) return MDVIT_E_SHAPE;
    snprintf(name, cap, "%s", g_samp_sym[index].name);
    *seen = g_samp_sym[index].seen; *timed = g_samp_sym[index].timed; *ms = g_samp_sym[index].ms;
    return MDVIT_OK;
}
static void sampler_note(const MdvitGemmDesc* d) {
    char nm[160];
    if (mdvit_gemm_kernel_name(d, nm, sizeof(nm)) != MDVIT_OK) return;
    if (char* plus = strchr(nm, '+')) *plus = 0;
    if (g_samp_only[0] && strcmp(g_samp_only, nm) != 0) return;
    int i = 0;
    while (i < g_samp_nsym && strcmp(g_samp_sym[i].name, nm) != 0) ++i;
    if (i == g_samp_nsym) {
        if (g_samp_nsym == SAMP_SYMS) return;
        snprintf(g_samp_sym[i].name, sizeof(g_samp_sym[i].name), "%s", nm);
        g_samp_sym[i].seen = 0; g_samp_sym[i].timed = 0; g_samp_sym[i].ms = 0.0;
        ++g_samp_nsym;
    }
    const long n = g_samp_sym[i].seen++;
    if (g_samp_n >= SAMP_MAX || (g_samp_stride > 1 && (((unsigned long)n * 2654435761UL) >> 7) % (unsigned long)g_samp_stride != 0)) return;
    if (g_samp_n >= g_samp_ev_made) {
        if (hipEventCreate(&g_samp_ev[g_samp_n][0]) != hipSuccess || hipEventCreate(&g_samp_ev[g_samp_n][1]) != hipSuccess) { (void)hipGetLastError(); return; }
        g_samp_ev_made = g_samp_n + 1;
    }
    g_samp_of[g_samp_n] = i;
    g_mdvit_t0 = g_samp_ev[g_samp_n][0]; g_mdvit_t1 = g_samp_ev[g_samp_n][1];          // consumed by this call's main-kernel launch (MDVIT_TIMED_LAUNCH)
    ++g_samp_n;
}

extern "C" int mdvit_gemm_f32(const MdvitGemmDesc* d, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    MDVIT_CHECK_ARG(d != nullptr, MDVIT_E_SHAPE, "gemm: null descriptor");
    MDVIT_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, MDVIT_E_SHAPE, "gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    if (g_ledger_on) ledger_note(d);
    if (g_samp_stride > 0 && d->A && d->B && d->C) sampler_note(d);
    MDVIT_CHECK_ARG(d->A && d->B && d->C, MDVIT_E_SHAPE, "gemm: null operand");
    MDVIT_CHECK_ARG(aligned16(d->A) && aligned16(d->B) && (d->lda % 4 == 0) && (d->ldb % 4 == 0), MDVIT_E_ALIGN,
                    "gemm: operands must be 16-byte aligned with leading dimensions %% 4 == 0 (lda=%ld ldb=%ld)", d->lda, d->ldb);
    MDVIT_CHECK_ARG(d->trans_a ? (d->M % 4 == 0) : (d->K % 4 == 0), MDVIT_E_ALIGN, "gemm: contiguous extent of A must be %% 4 (M=%d K=%d ta=%d)", d->M, d->K, d->trans_a);
    MDVIT_CHECK_ARG(d->trans_b ? (d->K % 4 == 0) : (d->N % 4 == 0), MDVIT_E_ALIGN, "gemm: contiguous extent of B must be %% 4 (N=%d K=%d tb=%d)", d->N, d->K, d->trans_b);
    MDVIT_CHECK_ARG(d->epi != MDVIT_EPI_DGELU || d->gelu_u || d->rc_a, MDVIT_E_SHAPE, "gemm: DGELU needs gelu_u (or rc_a/rc_b to recompute it)");
    if (d->rc_a) {
        MDVIT_CHECK_ARG(d->epi == MDVIT_EPI_DGELU && !d->gelu_u && d->rc_b && d->rc_k > 0 && !d->trans_a && d->trans_b, MDVIT_E_SHAPE,
                        "gemm: rc_a/rc_b (recomputed pre-activation) go with the NT DGELU epilogue and no gelu_u");
        MDVIT_CHECK_ARG(aligned16(d->rc_a) && aligned16(d->rc_b) && d->rc_lda % 4 == 0 && d->rc_ldb % 4 == 0 && d->rc_k % BK == 0 &&
                        (!d->rc_bias || aligned16(d->rc_bias)), MDVIT_E_ALIGN, "gemm: rc operands must be 16-byte aligned, leading dimensions %% 4 == 0, rc_k %% 32 == 0");
    }
    MDVIT_CHECK_ARG(!(d->e_drop_p > 0.f) || (long)d->M * d->N < (1L << 32), MDVIT_E_SHAPE, "gemm: dropout index space exceeds 2^32");

    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.A = d->A; a.B = d->B; a.C = d->C; a.C2 = d->C2;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.M = d->M; a.N = d->N; a.K = d->K;
    a.bias = d->bias;
    a.epi = d->epi;
    a.e_drop = d->e_drop_p > 0.f; a.e_k0 = d->e_key0; a.e_k1 = d->e_key1;
    a.e_thresh = mdvit_drop_thresh(d->e_drop_p); a.e_inv_keep = 1.f / (1.f - d->e_drop_p);
    a.e_rowscale = d->e_rowscale; a.e_rows_per_scale = d->e_rows_per_scale > 0 ? d->e_rows_per_scale : 1;
    a.residual = d->residual; a.ldr = d->ldr; a.gelu_u = d->gelu_u; a.ldu = d->ldu;
    a.rc_a = d->rc_a; a.rc_lda = d->rc_lda; a.rc_b = d->rc_b; a.rc_ldb = d->rc_ldb; a.rc_bias = d->rc_bias; a.rc_k = d->rc_k;
    a.accumulate = d->accumulate;
    a.colsum = d->colsum_a;
    MDVIT_CHECK_ARG(!d->colsum_a || (d->trans_a && !d->trans_b), MDVIT_E_SHAPE, "gemm: colsum_a rides on the TN (wgrad) layout only");
    a.seed = d->drop_seed;
    if (d->conv_c > 0) {
        MDVIT_CHECK_ARG(d->precision >= 1 && d->epi == MDVIT_EPI_NONE && !(d->e_drop_p > 0.f) && !d->e_rowscale && !d->residual && !d->rc_a, MDVIT_E_SHAPE,
                        "gemm: the implicit 3x3 convolution is built for precision >= 1 and the plain epilogue");
        MDVIT_CHECK_ARG(d->conv_c % BK == 0 && d->conv_h > 0 && d->conv_w > 0 && d->conv_ho > 0 && d->conv_wo > 0 && d->conv_stride >= 1 && d->conv_dilation >= 1,
                        MDVIT_E_SHAPE, "gemm: implicit convolution needs conv_c %% 32 == 0 and positive extents (conv_c=%d)", d->conv_c);
        if (d->trans_a && !d->trans_b) {       // weight gradient: B = the image, K = tokens, N = 9 conv_c (tap-major)
            MDVIT_CHECK_ARG(d->N == 9 * d->conv_c && d->K % (d->conv_ho * d->conv_wo) == 0 && mdvit_gemm_tn_applies(d), MDVIT_E_SHAPE,
                            "gemm: implicit-convolution weight gradient needs N == 9 conv_c, K == B conv_ho conv_wo (N=%d K=%d conv_c=%d)", d->N, d->K, d->conv_c);
        } else {
            MDVIT_CHECK_ARG(!d->trans_a && d->trans_b && d->K == 9 * d->conv_c && d->M % (d->conv_ho * d->conv_wo) == 0, MDVIT_E_SHAPE,
                            "gemm: implicit convolution (NT) needs K == 9 conv_c, M == B conv_ho conv_wo (M=%d K=%d conv_c=%d)", d->M, d->K, d->conv_c);
            a.conv_c = d->conv_c; a.conv_h = d->conv_h; a.conv_w = d->conv_w; a.conv_ho = d->conv_ho; a.conv_wo = d->conv_wo;
            a.conv_stride = d->conv_stride; a.conv_dil = d->conv_dilation; a.conv_up = d->conv_up > 1 ? d->conv_up : 1;
            MDVIT_CHECK_ARG(a.conv_up == 1 || d->conv_stride == 1, MDVIT_E_SHAPE, "gemm: conv_up (transposed convolution) goes with conv_stride == 1");
        }
    }
    if (mdvit_gemm_tn_applies(d)) return mdvit_gemm_tn_launch(d, s);
    MDVIT_CHECK_ARG(!d->a_bf16 && !d->b_bf16, MDVIT_E_SHAPE, "gemm: bf16-stored operands are a layout of the weight-gradient (TN, precision 1) kernel only");

    const GemmPlan pl = plan_gemm(d);
    a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n; a.splits = pl.splits; a.k_per_split = pl.kps;
    if (g_groups.n > 0) {
        MDVIT_CHECK_ARG(pl.splits == 1 && d->epi == MDVIT_EPI_NONE && !d->residual && !d->rc_a && d->conv_c <= 0 && !(d->e_drop_p > 0.f) && !d->e_rowscale,
                        MDVIT_E_SHAPE, "gemm (grouped): one K range, plain epilogue");
        a.ngroups = g_groups.n;
        for (int g = 0; g < g_groups.n; ++g) { a.gA[g] = g_groups.A[g]; a.gB[g] = g_groups.B[g]; a.gC[g] = g_groups.C[g]; a.gBias[g] = g_groups.bias[g]; }
    }
    if (pl.splits > 1) {
        const size_t need = sizeof(float) * (size_t)pl.splits * d->M * d->N;
        MDVIT_CHECK_ARG(d->ws != nullptr && d->ws_bytes >= need, MDVIT_E_WORKSPACE,
                        "gemm: split reduction needs %zu bytes of workspace (mdvit_gemm_ws_bytes), got %zu", need, (size_t)d->ws_bytes);
        a.slab = (float*)d->ws;
        a.bias = nullptr;                 // the reduce kernel adds the bias
    }
    // epilogue kind
    int epi = EPI_PLAIN;
    if (d->epi == MDVIT_EPI_GELU_DUAL) epi = EPI_GELU2;
    else if (d->epi == MDVIT_EPI_DGELU) epi = d->rc_a ? EPI_DGELU_RC : EPI_DGELU;
    else if (a.e_drop || d->e_rowscale || d->residual) epi = EPI_FULL;
    a.vec = ((d->N & 3) == 0) && ((d->ldc & 3) == 0) && aligned16(d->C);
    if (epi != EPI_PLAIN) {
        MDVIT_CHECK_ARG(a.vec && (!d->bias || aligned16(d->bias)) && (!d->C2 || aligned16(d->C2)) &&
                        (!d->residual || (aligned16(d->residual) && d->ldr % 4 == 0)) && (!d->gelu_u || (aligned16(d->gelu_u) && d->ldu % 4 == 0)),
                        MDVIT_E_ALIGN, "gemm: fused epilogues need N %% 4 == 0 and 16-byte aligned C / bias / residual / gelu_u rows");
        MDVIT_CHECK_ARG(!d->accumulate, MDVIT_E_SHAPE, "gemm: accumulate is only defined for the plain epilogue");
    } else if (d->bias && !aligned16(d->bias)) {
        a.vec = 0;
    }
    // stride-2 data gradient (conv_up == 2, even extents): tiles in parity-class order, dead taps skipped (gemm_body.inc); an internal schedule, same output rows
    bool conv_phase_on = d->conv_c > 0 && a.conv_up == 2;
    if (conv_phase_on) { const char* e = getenv("MDVIT_CONV_PHASE"); conv_phase_on = !(e && e[0] == '0'); }       // (read per call: the A/B switch of the parity test)
    a.conv_phase = (conv_phase_on && a.conv_ho == 2 * a.conv_h && a.conv_wo == 2 * a.conv_w && (d->M & 3) == 0 && (d->M >> 2) % 256 == 0 &&
                    pl.splits == 1 && a.vec && !d->accumulate) ? 1 : 0;
    int rc;
    if (pl.cfg == 0) rc = launch_cfg<128, 128, 2, 2>(a, d->trans_a, d->trans_b, epi, d->precision, s);
    else if (pl.cfg == 1) rc = launch_cfg<256, 64, 4, 1>(a, d->trans_a, d->trans_b, epi, d->precision, s);
    else rc = launch_cfg<64, 64, 2, 2>(a, d->trans_a, d->trans_b, epi, d->precision, s);
    MDVIT_CHECK_ARG(rc == 0, MDVIT_E_SHAPE, "gemm: this (trans_a=%d, trans_b=%d, epilogue=%d, precision=%d) combination is not built", d->trans_a, d->trans_b, epi, d->precision);
    if (pl.splits > 1) {
        const long total = (long)d->M * d->N / 4;
#define MDVIT_REDUCE_LAUNCH(R_) \
    hipLaunchKernelGGL((gemm_splitk_reduce_kernel<R_>), dim3((int)min((total * R_ + 255) / 256, 4096L)), dim3(256), 0, s, \
                       a.slab, d->bias, d->C, (long)d->ldc, d->M, d->N, pl.splits, d->accumulate, 0)
        if (total >= 65536 || pl.splits < 4) MDVIT_REDUCE_LAUNCH(1);
        else if (total >= 16384 || pl.splits < 16) MDVIT_REDUCE_LAUNCH(4);
        else if (total >= 4096 || pl.splits < 64) MDVIT_REDUCE_LAUNCH(16);
        else MDVIT_REDUCE_LAUNCH(64);
#undef MDVIT_REDUCE_LAUNCH
    }
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

// weight layouts of the implicit 3x3 convolution (see mdvit_hip.h)
__global__ __launch_bounds__(256) void conv_weight_relayout_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int mode) {
    const long total = (long)Cout * Cin * 9;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        if (mode == 0) {          // out[co][t][ci]
            const int ci = (int)(e % Cin); const int t = (int)((e / Cin) % 9); const int co = (int)(e / (9L * Cin));
            out[e] = w[((long)co * Cin + ci) * 9 + t];
        } else if (mode == 1) {   // out[ci][t][co] = w[co][ci][8 - t]
            const int co = (int)(e % Cout); const int t = (int)((e / Cout) % 9); const int ci = (int)(e / (9L * Cout));
            out[e] = w[((long)co * Cin + ci) * 9 + (8 - t)];
        } else {                  // back: out[co][ci][t] (+)= w[co][t][ci]   (mode 3 accumulates)
            const int t = (int)(e % 9); const int ci = (int)((e / 9) % Cin); const int co = (int)(e / (9L * Cin));
            const float v = w[((long)co * 9 + t) * Cin + ci];
            out[e] = mode == 3 ? out[e] + v : v;
        }
    }
}

// the same for many weights in ONE launch (modes 0 / 1 only): items [n][5] int64 in device memory = {w, out, Cout, Cin, mode}; grid.y = item.  Both modes move
// whole 128-byte lines on both sides through an LDS tile (round 5; the element-per-thread walk above read with a 36-byte stride and ran the step's ~40 layouts --
// 80 MB, alone on the GPU at the top of every forward -- at 0.5 TB/s: 158 us):
//   mode 0: tile = (co, 256 input channels): 2304 contiguous floats in, nine runs of 256 contiguous floats out
//   mode 1: tile = (32 output channels, 8 input channels): 32 runs of 72 contiguous floats in, 72 runs of 32 contiguous floats out
__global__ __launch_bounds__(256) void conv_weight_relayout_many_kernel(const long long* __restrict__ items) {
    __shared__ float s_t[32 * 73];
    const long long* it = items + 5 * (long)blockIdx.y;
    const float* w = reinterpret_cast<const float*>(it[0]);
    float* out = reinterpret_cast<float*>(it[1]);
    const int Cout = (int)it[2], Cin = (int)it[3], mode = (int)it[4];
    const int tid = threadIdx.x;
    if (mode == 0) {              // out[co][t][ci] = w[co][ci][t]
        const int nch = (Cin + 255) / 256, ntiles = Cout * nch;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int co = tile / nch, ci0 = (tile % nch) * 256, nci = min(256, Cin - ci0);
            const float* src = w + ((long)co * Cin + ci0) * 9;
            for (int i = tid; i < nci * 9; i += 256) s_t[i] = src[i];
            __syncthreads();
            for (int i = tid; i < nci * 9; i += 256) {
                const int t = i / nci, c = i - t * nci;
                out[((long)co * 9 + t) * Cin + ci0 + c] = s_t[c * 9 + t];
            }
            __syncthreads();
        }
    } else {                      // out[ci][t'][co] = w[co][ci][8 - t']
        const int ncb = (Cout + 31) / 32, nib = (Cin + 7) / 8, ntiles = ncb * nib;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int co0 = (tile / nib) * 32, ci0 = (tile % nib) * 8, ncl = min(32, Cout - co0), run = min(8, Cin - ci0) * 9;
            for (int i = tid; i < ncl * 72; i += 256) {
                const int c = i / 72, r = i - c * 72;
                if (r < run) s_t[c * 73 + r] = w[((long)(co0 + c) * Cin + ci0) * 9 + r];
            }
            __syncthreads();
            for (int i = tid; i < 72 * 32; i += 256) {
                const int c = i & 31, r = i >> 5, k = r / 9, tp = r - k * 9;
                if (c < ncl && k * 9 < run) out[((long)(ci0 + k) * 9 + tp) * Cout + co0 + c] = s_t[c * 73 + k * 9 + 8 - tp];
            }
            __syncthreads();
        }
    }
}

extern "C" int mdvit_conv_weight_relayout_many(const void* items_dev, int32_t n, int32_t blocks_per_item, void* stream) {
    MDVIT_CHECK_ARG(items_dev && n > 0 && blocks_per_item > 0, MDVIT_E_SHAPE, "conv_weight_relayout_many: bad arguments");
    hipLaunchKernelGGL(conv_weight_relayout_many_kernel, dim3(blocks_per_item, n), dim3(256), 0, (hipStream_t)stream, (const long long*)items_dev);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_conv_weight_relayout(const float* w, float* out, int32_t Cout, int32_t Cin, int32_t mode, void* stream) {
    MDVIT_CHECK_ARG(w && out && Cout > 0 && Cin > 0 && mode >= 0 && mode <= 3, MDVIT_E_SHAPE, "conv_weight_relayout: bad arguments");
    const long total = (long)Cout * Cin * 9;
    hipLaunchKernelGGL(conv_weight_relayout_kernel, dim3((int)min((total + 255) / 256, 2048L)), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, mode);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

// out[c][r] = in[r][c]  (weights only: the bf16x3 dgrad reads W^T so that both GEMM operands are k-contiguous)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, long ld_in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) out[(long)c * rows + r] = tile[tx][ty + 8 * i];
    }
}

// many small transposes in ONE launch: items [n][5] int64 in device memory = {in, out, ld_in, rows, cols}; grid.y = item
__global__ __launch_bounds__(256) void transpose_many_kernel(const long long* __restrict__ items) {
    __shared__ float tile[32][33];
    const long long* it = items + 5 * (long)blockIdx.y;
    const float* in = reinterpret_cast<const float*>(it[0]);
    float* out = reinterpret_cast<float*>(it[1]);
    const long ld_in = (long)it[2];
    const int rows = (int)it[3], cols = (int)it[4];
    const int tiles_x = (cols + 31) / 32, tiles_y = (rows + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tidx = blockIdx.x; tidx < tiles_x * tiles_y; tidx += gridDim.x) {
        const int r0 = (tidx / tiles_x) * 32, c0 = (tidx % tiles_x) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty + 8 * i, c = c0 + tx;
            tile[ty + 8 * i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + 8 * i, r = r0 + tx;
            if (c < cols && r < rows) out[(long)c * rows + r] = tile[tx][ty + 8 * i];
        }
        __syncthreads();
    }
}

extern "C" int mdvit_transpose_many(const void* items_dev, int32_t n, int32_t blocks_per_item, void* stream) {
    MDVIT_CHECK_ARG(items_dev && n > 0 && blocks_per_item > 0, MDVIT_E_SHAPE, "transpose_many: bad arguments");
    hipLaunchKernelGGL(transpose_many_kernel, dim3(blocks_per_item, n), dim3(256), 0, (hipStream_t)stream, (const long long*)items_dev);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

// up to 24 transposes in ONE launch with the items in the kernel ARGUMENTS (no device table: the operands are per-step temporaries -- the peer heads'
// composed weights, whose W^T every data-gradient product of both sweeps reads); grid.y = item, static-index select (no scratch)
namespace {
constexpr int TB_MAX = 24;
struct TransposeBatchArgs { const float* in[TB_MAX]; float* out[TB_MAX]; long ld[TB_MAX]; int rows[TB_MAX]; int cols[TB_MAX]; };
__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatchArgs a) {
    __shared__ float tile[32][33];
    const float* in = a.in[0]; float* out = a.out[0]; long ld_in = a.ld[0]; int rows = a.rows[0], cols = a.cols[0];
    const int y = blockIdx.y;
#pragma unroll
    for (int i = 1; i < TB_MAX; ++i)
        if (y == i) { in = a.in[i]; out = a.out[i]; ld_in = a.ld[i]; rows = a.rows[i]; cols = a.cols[i]; }
    const int tiles_x = (cols + 31) / 32, tiles_y = (rows + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tidx = blockIdx.x; tidx < tiles_x * tiles_y; tidx += gridDim.x) {
        const int r0 = (tidx / tiles_x) * 32, c0 = (tidx % tiles_x) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ty + 8 * i, c = c0 + tx;
            tile[ty + 8 * i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + 8 * i, r = r0 + tx;
            if (c < cols && r < rows) out[(long)c * rows + r] = tile[tx][ty + 8 * i];
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int mdvit_transpose_batch(int32_t n, const void* const* in, const int64_t* ld_in, void* const* out, const int32_t* rows, const int32_t* cols, void* stream) {
    MDVIT_CHECK_ARG(n >= 1 && n <= TB_MAX && in && ld_in && out && rows && cols, MDVIT_E_SHAPE, "transpose_batch: 1 <= n <= %d items", TB_MAX);
    TransposeBatchArgs a; memset(&a, 0, sizeof(a));
    int tiles = 1;
    for (int i = 0; i < n; ++i) {
        MDVIT_CHECK_ARG(in[i] && out[i] && rows[i] > 0 && cols[i] > 0 && ld_in[i] >= cols[i], MDVIT_E_SHAPE, "transpose_batch: item %d: bad shape", i);
        a.in[i] = (const float*)in[i]; a.out[i] = (float*)out[i]; a.ld[i] = (long)ld_in[i]; a.rows[i] = rows[i]; a.cols[i] = cols[i];
        tiles = max(tiles, cdiv(rows[i], 32) * cdiv(cols[i], 32));
    }
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(min(tiles, 64), n), dim3(256), 0, (hipStream_t)stream, a);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_transpose_f32(const float* in, int64_t ld_in, float* out, int32_t rows, int32_t cols, void* stream) {
    MDVIT_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld_in >= cols, MDVIT_E_SHAPE, "transpose: bad shape rows=%d cols=%d ld=%ld", rows, cols, (long)ld_in);
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, out, rows, cols);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_gemm_force_plan(int32_t cfg, int32_t splits) {
    g_force_cfg = (cfg >= 0 && cfg <= 2) ? cfg : -1;
    g_force_splits = splits > 0 ? splits : 0;
    return MDVIT_OK;
}

extern "C" int mdvit_gemm_plan(const MdvitGemmDesc* d, int32_t* tile_m, int32_t* tile_n, int32_t* splits) {
    MDVIT_CHECK_ARG(d != nullptr && d->M > 0 && d->N > 0 && d->K > 0, MDVIT_E_SHAPE, "gemm_plan: bad descriptor");
    if (mdvit_gemm_tn_applies(d)) { mdvit_gemm_tn_plan(d, tile_m, tile_n, splits); return MDVIT_OK; }
    const GemmPlan pl = plan_gemm(d);
    if (tile_m) *tile_m = pl.cfg == 0 ? 128 : (pl.cfg == 1 ? 256 : 64);
    if (tile_n) *tile_n = pl.cfg == 0 ? 128 : 64;
    if (splits) *splits = pl.splits;
    return MDVIT_OK;
}

// the kernel symbol this descriptor launches, as rocprofv3 prints it (bench.py matches its event timings against the profile by name)
void mdvit_gemm_tn_name(const MdvitGemmDesc* d, char* out, int cap);
extern "C" int mdvit_gemm_kernel_name(const MdvitGemmDesc* d, char* out, int32_t cap) {
    MDVIT_CHECK_ARG(d != nullptr && out != nullptr && cap > 0 && d->M > 0 && d->N > 0 && d->K > 0, MDVIT_E_SHAPE, "gemm_kernel_name: bad arguments");
    if (mdvit_gemm_tn_applies(d)) { mdvit_gemm_tn_name(d, out, cap); return MDVIT_OK; }
    const GemmPlan pl = plan_gemm(d);
    const bool drop = d->e_drop_p > 0.f;
    int epi = EPI_PLAIN;
    if (d->epi == MDVIT_EPI_GELU_DUAL) epi = EPI_GELU2;
    else if (d->epi == MDVIT_EPI_DGELU) epi = d->rc_a ? EPI_DGELU_RC : EPI_DGELU;
    else if (drop || d->e_rowscale || d->residual) epi = EPI_FULL;
    const int bm = pl.cfg == 0 ? 128 : (pl.cfg == 1 ? 256 : 64), bn = pl.cfg == 0 ? 128 : 64;
    if (d->conv_c > 0) {
        snprintf(out, cap, "gemm_conv3x3_kernel<%d, %d, %s>%s", bm, bn, pl.cfg == 1 ? "4, 1" : "2, 2", pl.splits > 1 ? "+splitk_reduce" : "");
        return MDVIT_OK;
    }
    snprintf(out, cap, "gemm_f32_kernel<%d, %d, %s, %s, %s, %d, %s>%s", bm, bn, pl.cfg == 1 ? "4, 1" : "2, 2", d->trans_a ? "true" : "false",
             d->trans_b ? "true" : "false", epi, d->precision ? "true" : "false", pl.splits > 1 ? "+splitk_reduce" : "");
    return MDVIT_OK;
}

extern "C" size_t mdvit_gemm_ws_bytes(const MdvitGemmDesc* d) {
    if (d == nullptr || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    if (mdvit_gemm_tn_applies(d)) return mdvit_gemm_tn_ws_bytes(d);
    const GemmPlan pl = plan_gemm(d);
    return pl.splits > 1 ? sizeof(float) * (size_t)pl.splits * d->M * d->N : 0;
}

// The same plain product on G operand triples in ONE launch (blockIdx.z = group): the peer heads' weight compositions (Decoders.py:315-339 through
// mdvit_amd/decode.py: W_fuse block x W_linear per head and scale) -- G x 4 small products that each paid a launch (and most a split-K reduction).
// desc: the shape / layout / precision of ONE group (its A / B / C are ignored); allow_split is forced off; plain epilogue, no bias.  NN / NT run the
// gemm.hip tiles, TN the weight-gradient kernel; per group the arithmetic is that of the single launch with allow_split = 0.
extern "C" int mdvit_gemm_f32_grouped(const MdvitGemmDesc* desc, int32_t G, const void* const* A, const void* const* B, void* const* C, void* stream) {
    MDVIT_CHECK_ARG(desc != nullptr && G >= 1 && G <= MDVIT_GEMM_MAX_GROUPS && A && B && C, MDVIT_E_SHAPE, "gemm (grouped): 1 <= G <= %d operand triples", MDVIT_GEMM_MAX_GROUPS);
    MdvitGemmDesc d = *desc;
    d.allow_split = 0; d.ws = nullptr; d.ws_bytes = 0;
    d.A = (const float*)A[0]; d.B = (const float*)B[0]; d.C = (float*)C[0];
    for (int g = 0; g < G; ++g) {
        MDVIT_CHECK_ARG(A[g] && B[g] && C[g] && aligned16(A[g]) && aligned16(B[g]), MDVIT_E_ALIGN, "gemm (grouped): group %d: null or unaligned operand", g);
        MDVIT_CHECK_ARG(aligned16(C[g]) == aligned16(C[0]), MDVIT_E_ALIGN, "gemm (grouped): the outputs must share their 16-byte alignment class");
        g_groups.A[g] = (const float*)A[g]; g_groups.B[g] = (const float*)B[g]; g_groups.C[g] = (float*)C[g];
    }
    for (int g = 0; g < G; ++g) g_groups.bias[g] = nullptr;
    g_groups.n = G;
    const int rc = mdvit_gemm_f32(&d, stream);
    g_groups.n = 0;
    return rc;
}

// ... with a bias vector per group (NN / NT only: the peer heads' low-resolution 1x1 convolutions, one launch over the G heads)
extern "C" int mdvit_gemm_f32_grouped_bias(const MdvitGemmDesc* desc, int32_t G, const void* const* A, const void* const* B, void* const* C, const void* const* bias,
                                           void* stream) {
    MDVIT_CHECK_ARG(desc != nullptr && G >= 1 && G <= MDVIT_GEMM_MAX_GROUPS && A && B && C && bias, MDVIT_E_SHAPE, "gemm (grouped): 1 <= G <= %d operand triples", MDVIT_GEMM_MAX_GROUPS);
    MDVIT_CHECK_ARG(!desc->trans_a, MDVIT_E_SHAPE, "gemm (grouped, bias): NN / NT only");
    MdvitGemmDesc d = *desc;
    d.allow_split = 0; d.ws = nullptr; d.ws_bytes = 0; d.bias = nullptr;
    d.A = (const float*)A[0]; d.B = (const float*)B[0]; d.C = (float*)C[0];
    for (int g = 0; g < G; ++g) {
        MDVIT_CHECK_ARG(A[g] && B[g] && C[g] && bias[g] && aligned16(A[g]) && aligned16(B[g]) && aligned16(bias[g]), MDVIT_E_ALIGN, "gemm (grouped): group %d: null or unaligned operand", g);
        MDVIT_CHECK_ARG(aligned16(C[g]) == aligned16(C[0]), MDVIT_E_ALIGN, "gemm (grouped): the outputs must share their 16-byte alignment class");
        g_groups.A[g] = (const float*)A[g]; g_groups.B[g] = (const float*)B[g]; g_groups.C[g] = (float*)C[g]; g_groups.bias[g] = (const float*)bias[g];
    }
    g_groups.n = G;
    const int rc = mdvit_gemm_f32(&d, stream);
    g_groups.n = 0;
    return rc;
}
