// Weight-gradient GEMM (TN): C[M,N] (+)= A^T B with A [K,M] and B [K,N] both TOKEN-major (k = token index, the long axis), fp32 in HBM,
// bf16x3 arithmetic (x = hi + lo bf16 planes; hi*lo + lo*hi + hi*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate) or one bf16 plane.
// Replaces autograd's `grad_out.t() @ input` of every nn.Linear / 1x1 nn.Conv2d of the reference (mdvit.py:288,310, mpvit.py:73,76,
// Decoders.py:185,300-311) in the backward of multi_train_MDViT.py:195-213.
//
// Both MFMA operands want 8 consecutive k per lane, but memory is contiguous along m / n.  The kernel keeps the k-major order all
// the way into LDS -- a float4 (four consecutive m at one k) is split and written as ONE ds_write_b64 per plane -- and transposes on the
// READ with gfx950's ds_read_b64_tr_b16: the LDS image is [k/4][m/16][4 k][16 m] bf16 blocks of 128 B; a 16-lane group hands the
// hardware the 16 eight-byte pieces of one block and every lane receives the 4 k values of "its" column m.  Two such reads give a
// lane its 8 k of one MFMA operand; the two 16-lane groups of a 32-lane half read ADJACENT blocks (256 contiguous bytes: no bank
// conflict), the 16 lanes of a write group fill one whole block (conflict-free too).
// K is split into slabs over blockIdx.y (dense [M,N] partial per slab + the fixed-order reduce of gemm.hip: deterministic);
// the loads of slab i+2 are in flight while slab i is multiplied (two register sets, two LDS stages, one barrier per slab).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef short v8i16 __attribute__((ext_vector_type(8)));

int mdvit_gemm_splitk_reduce_perm(const float* slab, const float* bias, float* C, long ldc, int M, int N, int splits, int accumulate, int perm_cin, hipStream_t s);
struct GemmGroups { int n; const float* A[MDVIT_GEMM_MAX_GROUPS]; const float* B[MDVIT_GEMM_MAX_GROUPS]; float* C[MDVIT_GEMM_MAX_GROUPS]; const float* bias[MDVIT_GEMM_MAX_GROUPS]; };
const GemmGroups* mdvit_gemm_groups_active();       // gemm.hip: the operand triples of a grouped launch in flight on this thread, or NULL

namespace {

constexpr int BK = 32, NTH = 256;

struct TnArgs {
    const float* A; const float* B; float* C; float* slab; float* colsum; float* cs_part; const float* bias;
    long lda, ldb, ldc;
    long slab_stride, cs_stride;                          // floats between the rows of two K-splits in slab / cs_part (M N and M; M N + M for both when the rows are joint)
    int M, N, K, kps, splits, tiles_m, tiles_n, accumulate, grid_xcd, perm_cin;
    int cv_c, cv_h, cv_w, cv_ho, cv_wo, cv_s, cv_d;      // CONVB: B is the NHWC image x, gathered as the im2col matrix [token][tap * C + c]
    int ngroups; const float* gA[MDVIT_GEMM_MAX_GROUPS]; const float* gB[MDVIT_GEMM_MAX_GROUPS]; float* gC[MDVIT_GEMM_MAX_GROUPS];      // grouped launch: blockIdx.z = group
#ifdef MDVIT_TN_PHASES
    long long* dbg;       // variant build (tools/probe/tn_phases.py): shader-cycle stamps of workgroup (tile 0, split 0), thread 0, [slab][8]
#endif
};
#ifdef MDVIT_TN_PHASES
static long long* g_tn_dbg = nullptr;
#define TNP(slab_, k_) do { if (p.dbg && tile == 0 && split == 0 && tid == 0 && (slab_) >= 0 && (slab_) < 64) p.dbg[(slab_) * 8 + (k_)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TNP(slab_, k_) do { } while (0)
#endif

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void split4(const float4 x, uint2& hi, uint2& lo) {
    f32x2_t a = {x.x, x.y}, b = {x.z, x.w};
    const uint32_t hau = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2_t));
    const uint32_t hbu = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bf16x2_t));
    f32x2_t la = {x.x - __uint_as_float(hau << 16), x.y - __uint_as_float(hau & 0xffff0000u)};
    f32x2_t lb = {x.z - __uint_as_float(hbu << 16), x.w - __uint_as_float(hbu & 0xffff0000u)};
    hi = make_uint2(hau, hbu);
    lo = make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(la, bf16x2_t)), __builtin_bit_cast(uint32_t, __builtin_convertvector(lb, bf16x2_t)));
}

typedef __attribute__((address_space(3))) v4i16* lds_v4i16_ptr;

// ABF / BBF: that operand is stored as bf16 in HBM ([K, M] / [K, N] of 2-byte elements, leading dimension in elements) -- the "mixed" mode's saved hidden
// activations of the C = 128 MLP (block.hip: store_bf16).  Its quads are loaded as 8 bytes, go to the hi plane unchanged, and its lo-plane MFMA is skipped.
template <int BM, int BN, int P, bool CS, bool CONVB, bool ABF = false, bool BBF = false>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(BM * BN == 16384 ? 2 : (BM * BN == 8192 ? 3 : 4), 8))) void gemm_tn_kernel(TnArgs p) {
    static_assert(!(CONVB && BBF) && (P == 2 || !(ABF || BBF)), "bf16-stored operands: plain TN products of the bf16x3 mode");
    constexpr int MB = BM / 16, NB = BN / 16;                     // 16-column blocks per k-quad
    constexpr int A_PLANE = BK * BM * 2, B_PLANE = BK * BN * 2;   // bytes of one bf16 plane of one slab
    constexpr int STAGE = P * (A_PLANE + B_PLANE);
    constexpr int AV = BM * BK / 4 / NTH, BV = BN * BK / 4 / NTH; // float4 per thread per slab
    constexpr int WTM = BM / 64, WTN = BN / 64;                   // 32x32 blocks per wave (2 x 2 waves)
    __shared__ __attribute__((aligned(256))) char smem[2 * STAGE];

    const float* gA = p.A; const float* gB = p.B;
    float* gC = p.C;
    if (p.ngroups > 0) {          // grouped launch (static indices: see gemm_body.inc)
        const int z = blockIdx.z;
#pragma unroll
        for (int g = 0; g < MDVIT_GEMM_MAX_GROUPS; ++g)
            if (z == g) { gA = p.gA[g]; gB = p.gB[g]; gC = p.gC[g]; }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt round robin to the 8 XCDs in launch order (x fastest).  The tiles of ONE K-split read the same token rows -- every A column block
    // once per tile column, every B column block once per tile row.  Remapping inside a split (tile) puts neighbouring tiles of a split on one XCD when a
    // split has many tiles; with <= 8 tiles per split it degenerates to "tile m of EVERY split on XCD m", and all eight L2s fetch all of the narrow operand.
    // grid_xcd (the launcher sets it for <= 8 tiles per split): the logical (split, tile) pair comes from the XCD-contiguous order of the WHOLE grid, so the
    // tiles of a split are neighbours on one XCD.  Measured (tools/tn_check.py --time): [262144 x 192]^T [262144 x 64] 67.9 -> 55.3 us, 64 x 512 134 -> 126,
    // 1024 x 128 over 65536 tokens 102 -> 98; with 16 tiles per split (512 x 512 over 4096 tokens) it loses (25.8 -> 29.7 us), hence the threshold.
    const int ntile = p.tiles_m * p.tiles_n;
    int tile, split;
    if (p.grid_xcd) {
        const int w = xcd_remap(blockIdx.y * ntile + blockIdx.x, ntile * p.splits);
        split = w / ntile; tile = w - split * ntile;
    } else {
        tile = xcd_remap(blockIdx.x, ntile); split = blockIdx.y;
    }
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * p.kps, kend = min(p.K, kbeg + p.kps);
    const int nslab = (kend - kbeg + BK - 1) / BK;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);
    const int l15 = lane & 15, l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging map: a wavefront covers one k-quad (4 rows) x 64 columns; lane -> (column quad mq = l&3, row kr = (l>>2)&3, block l>>4)
    const int kr = (lane >> 2) & 3;
    const int colq = 16 * (lane >> 4) + 4 * (lane & 3);           // column offset inside the 64-column unit
    float4 ra[2][AV], rb[2][BV];
    // CONVB (weight gradient of a 3x3 convolution, dW'[co][tap][ci] = sum_m dy[m][co] x[pixel(m) + tap][ci]): this thread's B columns
    // -- tap and channel -- are fixed for the whole K loop; per slab only the token (= output pixel) moves
    int bt_dy[CONVB ? BV : 1], bt_dx[CONVB ? BV : 1], bt_c[CONVB ? BV : 1];
    uint32_t bmask[2] = {0u, 0u};
    if (CONVB) {
#pragma unroll
        for (int v = 0; v < BV; ++v) {
            const int n = min(n0 + 64 * ((wave + 4 * v) >> 3) + colq, p.N - 4);
            const int tap = n / p.cv_c;
            bt_c[v] = n - tap * p.cv_c; bt_dy[v] = (tap / 3 - 1) * p.cv_d; bt_dx[v] = (tap % 3 - 1) * p.cv_d;
        }
    }
    float4 cs[BM / 64];                        // CS: the bias gradient (column sums of A) rides on the staging pass; one per 64-column unit
#pragma unroll
    for (int v = 0; v < BM / 64; ++v) cs[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_cs = CS && tn == 0;
    const int full_slabs = (kend - kbeg) / BK;                  // slabs at or past this index need their rows past kend zeroed

    // Row pointers of this thread's staged quads, stepped one slab at a time (round 4: the per-slab address chain -- clamp, 64-bit multiply by the leading
    // dimension, add -- was 10 quarter-rate v_mul_lo_u32 + 5 v_mad_u64_u32 per slab and wave, a fifth of a slab period's issue time on a kernel whose
    // waves run their phases back to back: tools/probe/tn_phases.py).  Full slabs read pa[v] + slab * BK * lda; the tail slab and the phantom
    // slabs behind it read the CLAMPED row of the tail slab (rows past kend are zeroed at the LDS store, as before).
    constexpr bool FASTA = !ABF, FASTB = !BBF && !CONVB;
    const float* pa[FASTA ? AV : 1]; long la[FASTA ? AV : 1];            // la / lb: the tail slab's clamped row MINUS the unclamped one (<= 0, elements)
    const float* pb[FASTB ? BV : 1]; long lb[FASTB ? BV : 1];
    if (FASTA) {
#pragma unroll
        for (int v = 0; v < AV; ++v) {
            const int u = wave + 4 * v, kk = 4 * (u & 7) + kr, m = min(m0 + 64 * (u >> 3) + colq, p.M - 4);
            pa[v] = gA + (long)(kbeg + kk) * p.lda + m;
            la[v] = (long)(min(kbeg + full_slabs * BK + kk, kend - 1) - (kbeg + full_slabs * BK + kk)) * p.lda;
        }
    }
    if (FASTB) {
#pragma unroll
        for (int v = 0; v < BV; ++v) {
            const int u = wave + 4 * v, kk = 4 * (u & 7) + kr, n = min(n0 + 64 * (u >> 3) + colq, p.N - 4);
            pb[v] = gB + (long)(kbeg + kk) * p.ldb + n;
            lb[v] = (long)(min(kbeg + full_slabs * BK + kk, kend - 1) - (kbeg + full_slabs * BK + kk)) * p.ldb;
        }
    }
    const long stepa = (long)BK * p.lda, stepb = (long)BK * p.ldb;
    // Loads are UNCONDITIONAL (addresses clamped into the matrix, rows past the end of the K range zeroed by a select): a load under
    // a branch makes the compiler drain the whole vector-memory queue (s_waitcnt vmcnt(0)) at every join, which serialises the
    // slabs.  Columns past M / N read real elements of the last column quad: they only reach output rows / columns that are never stored.
    auto load = [&](int slab, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        const int k0 = kbeg + slab * BK;
        // branch-free: the scalar slab offset (frozen at the tail slab) plus the per-thread clamp delta under a scalar all-ones / zero mask
        const long seff = (long)min(slab, full_slabs), tmask = slab >= full_slabs ? -1L : 0L;
#pragma unroll
        for (int v = 0; v < AV; ++v) {
            const int u = wave + 4 * v, k = k0 + 4 * (u & 7) + kr, m = min(m0 + 64 * (u >> 3) + colq, p.M - 4);
            if (ABF) {
                const uint2 b = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(gA) + (long)min(k, kend - 1) * p.lda + m);
                ra[S][v] = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), 0.f, 0.f);          // (bit containers)
            } else {
                ra[S][v] = *reinterpret_cast<const float4*>(pa[v] + seff * stepa + (la[v] & tmask));
            }
        }
        if (CONVB) {
            uint32_t ok = 0;
            const int hw = p.cv_ho * p.cv_wo;
#pragma unroll
            for (int v = 0; v < BV; ++v) {
                const int u = wave + 4 * v, m = min(k0 + 4 * (u & 7) + kr, kend - 1);
                const int bb = m / hw, rem = m - bb * hw, ho = rem / p.cv_wo, wo = rem - ho * p.cv_wo;
                const int y = ho * p.cv_s + bt_dy[v], x = wo * p.cv_s + bt_dx[v];
                const bool in = y >= 0 && y < p.cv_h && x >= 0 && x < p.cv_w;
                const int yc = min(max(y, 0), p.cv_h - 1), xc = min(max(x, 0), p.cv_w - 1);
                rb[S][v] = *reinterpret_cast<const float4*>(gB + ((long)(bb * p.cv_h + yc) * p.cv_w + xc) * p.cv_c + bt_c[v]);
                ok |= (in ? 1u : 0u) << v;
            }
            bmask[S] = ok;
        } else {
#pragma unroll
            for (int v = 0; v < BV; ++v) {
                const int u = wave + 4 * v, k = k0 + 4 * (u & 7) + kr, n = min(n0 + 64 * (u >> 3) + colq, p.N - 4);
                if (BBF) {
                    const uint2 b = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(gB) + (long)min(k, kend - 1) * p.ldb + n);
                    rb[S][v] = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), 0.f, 0.f);
                } else {
                    rb[S][v] = *reinterpret_cast<const float4*>(pb[v] + seff * stepb + (lb[v] & tmask));
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the loads HERE: the scheduler otherwise sinks them next to their LDS stores
    };
    auto store = [&](int slab, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        char* base = smem + (slab & 1) * STAGE;
        const int k0 = kbeg + slab * BK;
        if (slab >= full_slabs) {                           // (uniform: ONE scalar branch per slab; the per-quad tests under it run in the tail slab only)
#pragma unroll
            for (int v = 0; v < AV; ++v)
                if (k0 + 4 * ((wave + 4 * v) & 7) + kr >= kend) ra[S][v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int v = 0; v < BV; ++v)
                if (k0 + 4 * ((wave + 4 * v) & 7) + kr >= kend) rb[S][v] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#ifdef MDVIT_TN_PHASES
        TNP(slab - 1, 5);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AV + BV) : "memory");       // this set's loads have landed (the younger set's AV + BV may still fly)
        TNP(slab - 1, 6);
#endif
#pragma unroll
        for (int v = 0; v < AV; ++v) {
            const int u = wave + 4 * v;
            const int off = (((u & 7) * MB + 4 * (u >> 3) + (lane >> 4)) << 7) + l15 * 8;
            if (ABF) {
                const uint32_t b0 = __float_as_uint(ra[S][v].x), b1 = __float_as_uint(ra[S][v].y);
                if (CS) {
                    float4& c = cs[(4 * v) >> 3];
                    c.x += __uint_as_float(b0 << 16); c.y += __uint_as_float(b0 & 0xffff0000u); c.z += __uint_as_float(b1 << 16); c.w += __uint_as_float(b1 & 0xffff0000u);
                }
                *reinterpret_cast<uint2*>(base + off) = make_uint2(b0, b1);
                continue;
            }
            if (CS) { float4& c = cs[(4 * v) >> 3]; c.x += ra[S][v].x; c.y += ra[S][v].y; c.z += ra[S][v].z; c.w += ra[S][v].w; }
            uint2 hi, lo;
            split4(ra[S][v], hi, lo);
            *reinterpret_cast<uint2*>(base + off) = hi;
            if (P == 2) *reinterpret_cast<uint2*>(base + A_PLANE + off) = lo;
        }
        char* bb = base + P * A_PLANE;
#pragma unroll
        for (int v = 0; v < BV; ++v) {
            const int u = wave + 4 * v;
            const int off = (((u & 7) * NB + 4 * (u >> 3) + (lane >> 4)) << 7) + l15 * 8;
            if (CONVB && !((bmask[S] >> v) & 1u)) rb[S][v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BBF) {
                *reinterpret_cast<uint2*>(bb + off) = make_uint2(__float_as_uint(rb[S][v].x), __float_as_uint(rb[S][v].y));
                continue;
            }
            uint2 hi, lo;
            split4(rb[S][v], hi, lo);
            *reinterpret_cast<uint2*>(bb + off) = hi;
            if (P == 2) *reinterpret_cast<uint2*>(bb + B_PLANE + off) = lo;
        }
    };
    // fragment addresses (bytes inside a plane) of this lane for k-step 0, read 0, tile 0
    const int a_off = ((2 * lhi) * MB + wm0 / 16 + ((lane >> 4) & 1)) * 128 + l15 * 8;
    const int b_off = ((2 * lhi) * NB + wn0 / 16 + ((lane >> 4) & 1)) * 128 + l15 * 8;
    auto read8 = [&](const char* plane, int off, int blocks) __attribute__((always_inline)) -> bf16x8_t {
        // k-quads q and q+1 of this lane's half: two transposing reads, 4 k each
        const v4i16 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(plane + off));
        const v4i16 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_ptr)(plane + off + blocks * 128));
        const v8i16 r = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        return __builtin_bit_cast(bf16x8_t, r);
    };
    auto mma = [&](int stage) __attribute__((always_inline)) {
        const char* base = smem + stage * STAGE;
        const char* Ahi = base; const char* Alo = base + A_PLANE;
        const char* Bhi = base + P * A_PLANE; const char* Blo = Bhi + B_PLANE;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8_t ah[WTM], al[WTM], bh[WTN], bl[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                const int off = a_off + (4 * ks * MB + 2 * i) * 128;
                ah[i] = read8(Ahi, off, MB);
                if (P == 2 && !ABF) al[i] = read8(Alo, off, MB);
            }
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int off = b_off + (4 * ks * NB + 2 * j) * 128;
                bh[j] = read8(Bhi, off, NB);
                if (P == 2 && !BBF) bl[j] = read8(Blo, off, NB);
            }
            if (P == 2 && !BBF) {
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            }
            if (P == 2 && !ABF) {
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // branch-free pipeline over an EVEN number of slabs (an odd count gets one phantom slab of zeros): slab i in LDS stage i & 1,
    // the loads of slab i+2 in flight while slab i is multiplied and slab i+1 is split into the other stage; one barrier per slab
    load(0, S0{});
    load(1, S1{});
    store(0, S0{});
    __syncthreads();
    for (int i = 0; i < nslab; i += 2) {
        TNP(i, 0);
        load(i + 2, S0{});
        TNP(i, 1);
        mma(0);
        TNP(i, 2);
        store(i + 1, S1{});
        TNP(i, 3);
        __syncthreads();
        TNP(i, 4);
        load(i + 3, S1{});
        TNP(i + 1, 1);
        mma(1);
        TNP(i + 1, 2);
        store(i + 2, S0{});
        TNP(i + 1, 3);
        __syncthreads();
        TNP(i + 1, 4);
    }

    if (CS && do_cs) {              // bias gradient: column sums of the A stream, added in a FIXED order (the loop ended with a barrier)
        float* s_cs = reinterpret_cast<float*>(smem);               // [4 waves][BM]
#pragma unroll
        for (int h = 0; h < BM / 64; ++h) {
            float4 c = cs[h];                                        // this thread: rows kr of the wave's k-quads; fold the 4 row lanes
            c.x += __shfl_xor(c.x, 4); c.y += __shfl_xor(c.y, 4); c.z += __shfl_xor(c.z, 4); c.w += __shfl_xor(c.w, 4);
            c.x += __shfl_xor(c.x, 8); c.y += __shfl_xor(c.y, 8); c.z += __shfl_xor(c.z, 8); c.w += __shfl_xor(c.w, 8);
            if (kr == 0) *reinterpret_cast<float4*>(&s_cs[wave * BM + 64 * h + colq]) = c;
        }
        __syncthreads();
        for (int i = tid; i < BM; i += NTH) {
            if (m0 + i >= p.M) continue;
            const float t = (s_cs[i] + s_cs[BM + i]) + (s_cs[2 * BM + i] + s_cs[3 * BM + i]);
            if (p.splits > 1) p.cs_part[(long)split * p.cs_stride + m0 + i] = t;     // one row per K-split, summed by the slab reduction
            else p.colsum[m0 + i] += t;                                          // (this workgroup is the only writer of these columns)
        }
    }

    // epilogue: the MFMA ran as D = B-tile^T x A-tile -> D[row = n][col = m]; a lane holds four consecutive n per register quad
    float* slab = p.splits > 1 ? p.slab + (long)split * p.slab_stride : nullptr;
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
        const int row = m0 + wm0 + i * 32 + l31;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + wn0 + j * 32 + 8 * q + 4 * lhi;
                if (col >= p.N) continue;
                float4 v = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (slab) { *reinterpret_cast<float4*>(slab + (long)row * p.N + col) = v; continue; }
                if (p.perm_cin > 0) {          // conv weight gradient straight into the [Cout, Cin, 3, 3] layout (see mdvit_gemm_splitk_reduce_perm)
                    const float a4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const int nn = col + t4, tap = nn / p.perm_cin, ci = nn - tap * p.perm_cin;
                        float* d1 = gC + (long)row * p.ldc + (long)ci * 9 + tap;
                        *d1 = p.accumulate ? *d1 + a4[t4] : a4[t4];
                    }
                    continue;
                }
                float* dst = gC + (long)row * p.ldc + col;
                if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *reinterpret_cast<float4*>(dst) = v;
            }
    }
}

int g_tn_enable = 1, g_tn_force_cfg = -1, g_tn_force_splits = 0;
int g_tn_grid_xcd = -1;          // -1: from MDVIT_TN_GRID_XCD at the first launch (unset / 1: whole-grid XCD order for <= 8 tiles per split; 0: never; 2: always)

struct TnPlan { int cfg, tiles_m, tiles_n, splits, kps; };
const int TN_BM[4] = {128, 128, 64, 64}, TN_BN[4] = {128, 64, 128, 64};

TnPlan plan_tn(int M, int N, int K, int allow_split) {
    static const double EFF[4] = {1.0, 0.9, 0.9, 0.75};
    TnPlan best; best.cfg = 0; double best_cost = 1e300;
    for (int c = 0; c < 4; ++c) {
        if (g_tn_force_cfg >= 0 && c != g_tn_force_cfg) continue;
        const long tm = cdiv(M, TN_BM[c]), tn = cdiv(N, TN_BN[c]);
        const double cost = (double)(tm * TN_BM[c]) * (double)(tn * TN_BN[c]) / EFF[c];
        if (cost < best_cost) { best_cost = cost; best.cfg = c; best.tiles_m = (int)tm; best.tiles_n = (int)tn; }
    }
    const long tiles = (long)best.tiles_m * best.tiles_n;
    // Workgroups in total: ~1.5 per CU (384).  ALONE on the GPU the kernel prefers a whole multiple of the 256 CUs from below (tools/tn_plan_sweep.py, every
    // shape of the model x every tile x 128 .. 1024 workgroups: minima at <= 256 and <= 512, 320 - 384 are 8 - 15 % slower -- the CUs that get a second
    // workgroup set the kernel's time), but the weight gradients run on the side stream NEXT to the data-gradient chain, and there the rule that wins alone
    // loses: three interleaved A/B pairs of the bench step, 384: 431.7 / 430.5 / 430.4 images/s, CU multiples: 429.2 / 427.7 / 428.7
    // (MDVIT_TN_WORKGROUPS=-1 selects the CU-multiple rule, N > 0 another total: A/B hook).  At least 8 slabs of 32 tokens each per workgroup.
    static const int total_env = [] { const char* e = getenv("MDVIT_TN_WORKGROUPS"); return e ? atoi(e) : 0; }();
    long want;
    if (total_env < 0) {
        const long target = (best.cfg == 0 || tiles == 1) ? 256 : 512;       // 128 x 128 tiles (two fit a CU) and single-tile outputs: one per CU; smaller tiles: two
        want = target / tiles;
        if (want * tiles * 4 < target * 3) want = 2 * target / tiles;
        if (want < 1) want = 1;
    } else {
        const long total = total_env > 0 ? total_env : 384;
        want = (total + tiles - 1) / tiles;
    }
    if (g_tn_force_splits > 0) want = g_tn_force_splits;
    const long max_sp = K / (8 * BK) > 0 ? K / (8 * BK) : 1;
    if (want > max_sp) want = max_sp;
    if (want > 1024) want = 1024;
    if (want < 1 || !allow_split) want = 1;
    best.kps = cdiv(cdiv(K, want), BK) * BK;
    best.splits = cdiv(K, best.kps);
    return best;
}

}  // namespace

// library-internal: does the transposing-read wgrad kernel take this descriptor?
bool mdvit_gemm_tn_applies(const MdvitGemmDesc* d) {
    return g_tn_enable && d->trans_a && !d->trans_b && d->precision >= 1 && d->epi == MDVIT_EPI_NONE && !(d->e_drop_p > 0.f) && !d->e_rowscale &&
           !d->residual && (d->M % 4 == 0) && (d->N % 4 == 0) && (d->ldc % 4 == 0) && aligned16(d->C) && (!d->bias || aligned16(d->bias));
}

size_t mdvit_gemm_tn_ws_bytes(const MdvitGemmDesc* d) {
    const TnPlan pl = plan_tn(d->M, d->N, d->K, d->allow_split);
    return pl.splits > 1 ? sizeof(float) * (size_t)pl.splits * d->M * (d->N + (d->colsum_a ? 1 : 0)) : 0;
}

void mdvit_gemm_tn_plan(const MdvitGemmDesc* d, int* tile_m, int* tile_n, int* splits) {
    const TnPlan pl = plan_tn(d->M, d->N, d->K, d->allow_split);
    if (tile_m) *tile_m = TN_BM[pl.cfg];
    if (tile_n) *tile_n = TN_BN[pl.cfg];
    if (splits) *splits = pl.splits;
}

void mdvit_gemm_tn_name(const MdvitGemmDesc* d, char* out, int cap) {
    const TnPlan pl = plan_tn(d->M, d->N, d->K, d->allow_split);
    if (d->a_bf16 || d->b_bf16) {
        snprintf(out, cap, "gemm_tn_kernel<%d, %d, 2, %s, false, %s, %s>%s", TN_BM[pl.cfg], TN_BN[pl.cfg], d->colsum_a ? "true" : "false", d->a_bf16 ? "true" : "false",
                 d->b_bf16 ? "true" : "false", pl.splits > 1 ? "+splitk_reduce" : "");
        return;
    }
    // (all seven template arguments, as rocprofv3 prints the symbol: bench.py matches this name against the committed kernel-trace summaries)
    snprintf(out, cap, "gemm_tn_kernel<%d, %d, %d, %s, %s, false, false>%s", TN_BM[pl.cfg], TN_BN[pl.cfg], (d->precision == 2 && d->conv_c <= 0) ? 1 : 2,
             d->colsum_a ? "true" : "false", d->conv_c > 0 ? "true" : "false", pl.splits > 1 ? "+splitk_reduce" : "");
}

int mdvit_gemm_tn_launch(const MdvitGemmDesc* d, hipStream_t s) {
    if (d->a_bf16 || d->b_bf16)
        MDVIT_CHECK_ARG(!(d->a_bf16 && d->b_bf16) && d->conv_c <= 0 && d->precision == 1 && d->lda % 4 == 0 && d->ldb % 4 == 0, MDVIT_E_SHAPE,
                        "gemm (wgrad): a bf16-stored operand needs precision 1, no convolution, the other operand in fp32, leading dimensions %% 4 == 0");
    const TnPlan pl = plan_tn(d->M, d->N, d->K, d->allow_split);
    TnArgs a;
    memset(&a, 0, sizeof(a));
    a.A = d->A; a.B = d->B; a.C = d->C; a.colsum = d->colsum_a; a.bias = d->bias;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.M = d->M; a.N = d->N; a.K = d->K;
    a.kps = pl.kps; a.splits = pl.splits; a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n; a.accumulate = d->accumulate;
    if (g_tn_grid_xcd < 0) { const char* e = getenv("MDVIT_TN_GRID_XCD"); g_tn_grid_xcd = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }
#ifdef MDVIT_TN_PHASES
    a.dbg = g_tn_dbg;
#endif
    a.grid_xcd = g_tn_grid_xcd == 2 || (g_tn_grid_xcd == 1 && pl.tiles_m * pl.tiles_n <= 8 && pl.splits > 1);
    if (d->conv_c > 0 && d->conv_wgrad_nchw) {
        MDVIT_CHECK_ARG(!d->bias && d->N == 9 * d->conv_c, MDVIT_E_SHAPE, "gemm (wgrad): conv_wgrad_nchw needs N == 9 conv_c and no bias");
        a.perm_cin = d->conv_c;
    }
    if (d->conv_c > 0) {
        a.cv_c = d->conv_c; a.cv_h = d->conv_h; a.cv_w = d->conv_w; a.cv_ho = d->conv_ho; a.cv_wo = d->conv_wo; a.cv_s = d->conv_stride; a.cv_d = d->conv_dilation;
    }
    if (pl.splits > 1) {
        const size_t need = sizeof(float) * (size_t)pl.splits * d->M * (d->N + (d->colsum_a ? 1 : 0));
        MDVIT_CHECK_ARG(d->ws != nullptr && d->ws_bytes >= need, MDVIT_E_WORKSPACE,
                        "gemm (wgrad): split reduction needs %zu bytes of workspace (mdvit_gemm_ws_bytes), got %zu", need, (size_t)d->ws_bytes);
        a.slab = (float*)d->ws;
        a.cs_part = a.slab + (size_t)pl.splits * d->M * d->N;
        a.slab_stride = (long)d->M * d->N; a.cs_stride = d->M;
        // (Round 5 tried joint rows -- a K-split's product and its column sums side by side, [splits][M N + M], ONE launch of the partial-row kernel for dW and db -- and the
        // many-split reductions of the small outputs through that kernel as well: 63 launches per step fewer on the weight-gradient stream, but the partial-row kernels
        // are slower than gemm_splitk_reduce on wide outputs: 4.36 against 3.78 ms of reduction kernels per bs=4 step (rocprofv3, profiles/r05d / r05c kernel stats), the
        // step unchanged in three interleaved A/B pairs.  Not kept; slab_stride / cs_stride stay as kernel arguments.)
    }
    if (const GemmGroups* gg = mdvit_gemm_groups_active()) {
        MDVIT_CHECK_ARG(pl.splits == 1 && !d->colsum_a && d->conv_c <= 0 && !d->a_bf16 && !d->b_bf16, MDVIT_E_SHAPE, "gemm (wgrad, grouped): one K range, no column sums, no gather");
        a.ngroups = gg->n;
        for (int g = 0; g < gg->n; ++g) { a.gA[g] = gg->A[g]; a.gB[g] = gg->B[g]; a.gC[g] = gg->C[g]; }
    }
    const dim3 grid(pl.tiles_m * pl.tiles_n, pl.splits, a.ngroups > 0 ? a.ngroups : 1), block(NTH);
    // (MDVIT_EXP_TN_ONE_PLANE=1: an EXPERIMENT switch, not a mode -- every weight-gradient product on one bf16 plane, to measure how much of the step a faster
    //  weight-gradient kernel could buy: tools/experiments/README.md)
    static const bool exp_one = [] { const char* e = getenv("MDVIT_EXP_TN_ONE_PLANE"); return e && e[0] == '1'; }();
    const bool one = d->precision == 2 || (exp_one && !d->a_bf16 && !d->b_bf16);
#define MDVIT_TN_LAUNCH(BM_, BN_)                                                                                   \
    do {                                                                                                            \
        if (a.cv_c > 0) {                                                                                           \
            if (a.colsum) MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, true, true>), grid, block, 0, s, a);      \
            else MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, false, true>), grid, block, 0, s, a);              \
        } else if (d->a_bf16) {                                                                                     \
            if (a.colsum) MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, true, false, true, false>), grid, block, 0, s, a);      \
            else MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, false, false, true, false>), grid, block, 0, s, a);              \
        } else if (d->b_bf16) {                                                                                     \
            if (a.colsum) MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, true, false, false, true>), grid, block, 0, s, a);      \
            else MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, false, false, false, true>), grid, block, 0, s, a);              \
        } else if (one) { if (a.colsum) MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 1, true, false>), grid, block, 0, s, a);     \
                   else MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 1, false, false>), grid, block, 0, s, a); }           \
        else { if (a.colsum) MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, true, false>), grid, block, 0, s, a);         \
               else MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, false, false>), grid, block, 0, s, a); }               \
    } while (0)
    if (pl.cfg == 0) MDVIT_TN_LAUNCH(128, 128);
    else if (pl.cfg == 1) MDVIT_TN_LAUNCH(128, 64);
    else if (pl.cfg == 2) MDVIT_TN_LAUNCH(64, 128);
    else MDVIT_TN_LAUNCH(64, 64);
#undef MDVIT_TN_LAUNCH
    MDVIT_LAUNCH_CHECK();
    if (pl.splits > 1) {
        int rc = mdvit_gemm_splitk_reduce_perm(a.slab, d->bias, d->C, d->ldc, d->M, d->N, pl.splits, d->accumulate, a.perm_cin, s);
        if (rc == MDVIT_OK && d->colsum_a)        // the K-splits' column-sum rows, added in slab order
            rc = mdvit_reduce_partials(a.cs_part, pl.splits, d->M, d->M, d->colsum_a, 0, nullptr, 1, s);
        return rc;
    }
    return MDVIT_OK;
}

extern "C" int mdvit_gemm_tn_grid_order(int32_t mode) {
    g_tn_grid_xcd = (mode >= 0 && mode <= 2) ? mode : -1;
    return MDVIT_OK;
}

extern "C" int mdvit_gemm_tn_config(int32_t enable, int32_t cfg, int32_t splits) {
    g_tn_enable = enable != 0;
    g_tn_force_cfg = (cfg >= 0 && cfg <= 3) ? cfg : -1;
    g_tn_force_splits = splits > 0 ? splits : 0;
    return MDVIT_OK;
}

#ifdef MDVIT_TN_PHASES
extern "C" void mdvit_tn_debug_buffer(long long* buf) { g_tn_dbg = buf; }
#endif
