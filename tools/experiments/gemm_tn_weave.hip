// EXPERIMENT (round 3; measured, not kept): gemm_tn.hip with (a) FAST loads for the full slabs -- a uniform base pointer stepping one slab at a time plus a
// constant per-thread 32-bit byte offset, no row clamp, no tail selects (the address arithmetic was 92 of the 221 VALU instructions of a slab period) -- and
// (b) the split of slab i+1 WOVEN into the MFMAs of slab i in a fixed order (scheduling barriers MFMA by MFMA; inline-asm pins keep instruction selection
// from splitting a slab in the half period that loads it).  The ISA comes out as designed (per k-step: 16 fragment reads, then 3 MFMA | 12 VALU | ...), VGPRs
// 228 -> 250-256, and the kernel is 2-5 % faster on the large shapes, equal on the small ones (tools/tn_check.py --time, one MI355X):
//     [65536 x 1024]^T [65536 x 128]  102.0 us (104.4)      [16384 x 1280]^T [16384 x 320]  80.5 (84.4)      [262144 x 192]^T [262144 x 64]  66.0 (69.9)
// One bf16 plane instead of three MFMAs per product moves the same shapes by < 5 % too: a slab period takes ~4900 cycles next to ~770 cycles of MFMA and
// ~900 of VALU -- and it is not memory either (tools/probe/tn_cached_operands_probe.py: cache-resident operands, same time).  SQ counters: 45 % of a wave's
// cycles wait for instruction results, 16 % at barriers, with one or two waves per SIMD (profiles/r03_pmc_tn_stalls.txt).  Not adopted: 3 % of a kernel family
// that is 9 % of the step's kernel time.  With the split's packed subtractions written as scalar v_sub_f32 and -fno-slp-vectorize (beside MFMAs a v_pk_add_f32
// costs ~13 cycles more than the pair it replaces): 1280 x 320 over 16384 tokens 77.8 us (84.9), 960 x 320 66.4 (71.1), 1024 x 128 over 65536 94.9 (96.5).
// Build: replace mdvit_amd/csrc/gemm_tn.hip with this file (same entry points).
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

int mdvit_gemm_splitk_reduce(const float* slab, const float* bias, float* C, long ldc, int M, int N, int splits, int accumulate, hipStream_t s);

namespace {

constexpr int BK = 32, NTH = 256;

struct TnArgs {
    const float* A; const float* B; float* C; float* slab; float* colsum; float* cs_part; const float* bias;
    long lda, ldb, ldc;
    int M, N, K, kps, splits, tiles_m, tiles_n, accumulate, grid_xcd;
    int cv_c, cv_h, cv_w, cv_ho, cv_wo, cv_s, cv_d;      // CONVB: B is the NHWC image x, gathered as the im2col matrix [token][tap * C + c]
};

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

template <int BM, int BN, int P, bool CS, bool CONVB>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu(BM * BN == 16384 ? 2 : (BM * BN == 8192 ? 3 : 4), 8))) void gemm_tn_kernel(TnArgs p) {
    constexpr int MB = BM / 16, NB = BN / 16;                     // 16-column blocks per k-quad
    constexpr int A_PLANE = BK * BM * 2, B_PLANE = BK * BN * 2;   // bytes of one bf16 plane of one slab
    constexpr int STAGE = P * (A_PLANE + B_PLANE);
    constexpr int AV = BM * BK / 4 / NTH, BV = BN * BK / 4 / NTH; // float4 per thread per slab
    constexpr int WTM = BM / 64, WTN = BN / 64;                   // 32x32 blocks per wave (2 x 2 waves)
    __shared__ __attribute__((aligned(256))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt round robin to the 8 XCDs in launch order (x fastest).  The tiles of ONE K-split read the same token rows -- every A column block
    // once per tile column, every B column block once per tile row -- so they belong on one XCD, next to each other in time: the logical (split, tile) pair
    // is taken from the XCD-contiguous order of the WHOLE grid.  (Remapping inside a split only, as before, put tile m of every split on XCD m whenever a
    // split has <= 8 tiles: all eight L2s fetched all of the narrow operand -- 537 MB for the 302 MB of a [65536 x 1024]^T [65536 x 128] product.)
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

    // Loads are UNCONDITIONAL (addresses clamped into the matrix, rows past the end of the K range zeroed by a select): a load under
    // a branch makes the compiler drain the whole vector-memory queue (s_waitcnt vmcnt(0)) at every join, which serialises the
    // slabs.  Columns past M / N read real elements of the last column quad: they only reach output rows / columns that are never stored.
    //
    // FAST loads (slabs wholly inside [kbeg, kend), i.e. all but the last one or two of a workgroup): a UNIFORM base pointer that steps one slab at a time
    // plus a per-thread 32-bit byte offset that never changes -- no vector address arithmetic in the loop.  Computing the address from the slab index
    // (64-bit multiply-add per float4, row clamp) was 92 of the 221 VALU instructions of a slab period, the quarter-rate v_mul_lo_u32 among them, next to
    // 24 MFMAs of 32 cycles each: the period was VALU-bound in the wave.
    uint32_t offa[AV], offb[CONVB ? 1 : BV];
#pragma unroll
    for (int v = 0; v < AV; ++v) {
        const int u = wave + 4 * v, m = min(m0 + 64 * (u >> 3) + colq, p.M - 4);
        offa[v] = (uint32_t)(((long)(4 * (u & 7) + kr) * p.lda + m) * 4);
    }
    if (!CONVB) {
#pragma unroll
        for (int v = 0; v < BV; ++v) {
            const int u = wave + 4 * v, n = min(n0 + 64 * (u >> 3) + colq, p.N - 4);
            offb[v] = (uint32_t)(((long)(4 * (u & 7) + kr) * p.ldb + n) * 4);
        }
    }
    const char* abase = reinterpret_cast<const char*>(p.A + (long)(kbeg + 2 * BK) * p.lda);      // slab 2: the first FAST load of the loop
    const char* bbase = reinterpret_cast<const char*>(p.B + (long)(kbeg + 2 * BK) * p.ldb);
    const long a_step = (long)BK * p.lda * 4, b_step = (long)BK * p.ldb * 4;
    auto load = [&](int slab, auto setc, auto fastc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        constexpr bool FAST = decltype(fastc)::value;
        const int k0 = kbeg + slab * BK;
#pragma unroll
        for (int v = 0; v < AV; ++v) {
            if (FAST) {
                ra[S][v] = *reinterpret_cast<const float4*>(abase + offa[v]);
            } else {
                const int u = wave + 4 * v, k = k0 + 4 * (u & 7) + kr, m = min(m0 + 64 * (u >> 3) + colq, p.M - 4);
                ra[S][v] = *reinterpret_cast<const float4*>(p.A + (long)min(k, kend - 1) * p.lda + m);
            }
        }
        if (FAST) abase += a_step;
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
                rb[S][v] = *reinterpret_cast<const float4*>(p.B + ((long)(bb * p.cv_h + yc) * p.cv_w + xc) * p.cv_c + bt_c[v]);
                ok |= (in ? 1u : 0u) << v;
            }
            bmask[S] = ok;
        } else {
#pragma unroll
            for (int v = 0; v < BV; ++v) {
                if (FAST) {
                    rb[S][v] = *reinterpret_cast<const float4*>(bbase + offb[v]);
                } else {
                    const int u = wave + 4 * v, k = k0 + 4 * (u & 7) + kr, n = min(n0 + 64 * (u >> 3) + colq, p.N - 4);
                    rb[S][v] = *reinterpret_cast<const float4*>(p.B + (long)min(k, kend - 1) * p.ldb + n);
                }
            }
            if (FAST) bbase += b_step;
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the loads HERE: the scheduler otherwise sinks them next to their LDS stores
    };
    // one staged float4 (chunk c: A quads first, then B quads) of register set S: zero what lies outside, column sums, split into the two bf16 planes
    constexpr int NCH = AV + BV;
    uint2 shi[NCH], slo[NCH];
    auto split_chunk = [&](int c, int slab, auto setc, auto fastc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        const bool tail = !decltype(fastc)::value && slab >= full_slabs;               // (uniform; FAST: a full slab by construction, no row selects)
        const int k0 = kbeg + slab * BK;
        // the staged quad becomes "defined" HERE: the split is pure arithmetic, and without the pin instruction selection places it right behind the load
        // (every wave then sits on the latency of the load it has just issued)
        if (c < AV) asm volatile("" : "+v"(ra[S][c].x), "+v"(ra[S][c].y), "+v"(ra[S][c].z), "+v"(ra[S][c].w));
        else asm volatile("" : "+v"(rb[S][c - AV].x), "+v"(rb[S][c - AV].y), "+v"(rb[S][c - AV].z), "+v"(rb[S][c - AV].w));
        if (c < AV) {
            const int v = c, u = wave + 4 * v;
            if (tail && k0 + 4 * (u & 7) + kr >= kend) ra[S][v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (CS) { float4& q = cs[(4 * v) >> 3]; q.x += ra[S][v].x; q.y += ra[S][v].y; q.z += ra[S][v].z; q.w += ra[S][v].w; }
            split4(ra[S][v], shi[c], slo[c]);
        } else {
            const int v = c - AV, u = wave + 4 * v;
            if (tail && k0 + 4 * (u & 7) + kr >= kend) rb[S][v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (CONVB && !((bmask[S] >> v) & 1u)) rb[S][v] = make_float4(0.f, 0.f, 0.f, 0.f);
            split4(rb[S][v], shi[c], slo[c]);
        }
    };
    auto write_chunk = [&](int c, int slab) __attribute__((always_inline)) {
        char* base = smem + (slab & 1) * STAGE;
        if (c < AV) {
            const int u = wave + 4 * c;
            const int off = (((u & 7) * MB + 4 * (u >> 3) + (lane >> 4)) << 7) + l15 * 8;
            *reinterpret_cast<uint2*>(base + off) = shi[c];
            if (P == 2) *reinterpret_cast<uint2*>(base + A_PLANE + off) = slo[c];
        } else {
            const int u = wave + 4 * (c - AV);
            const int off = (((u & 7) * NB + 4 * (u >> 3) + (lane >> 4)) << 7) + l15 * 8;
            char* bb = base + P * A_PLANE;
            *reinterpret_cast<uint2*>(bb + off) = shi[c];
            if (P == 2) *reinterpret_cast<uint2*>(bb + B_PLANE + off) = slo[c];
        }
    };
    auto store = [&](int slab, auto setc) __attribute__((always_inline)) {      // (prologue only)
        using SLOWT = std::integral_constant<bool, false>;
#pragma unroll
        for (int c = 0; c < NCH; ++c) { split_chunk(c, slab, setc, SLOWT{}); write_chunk(c, slab); }
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
    // One slab period: the MFMAs of the slab in LDS stage `stage` WOVEN with the split of the next slab (register set S, LDS stage `stage ^ 1`).  A wave can
    // issue VALU work while its own MFMA runs its 8 passes, and with ~1.5 workgroups per CU there is often no second wave on the SIMD to fill either phase:
    // the order is fixed here, MFMA by MFMA (scheduling barriers), with the NCH float4 splits spread evenly over the 2 x NM MFMAs.  The LDS stores of the
    // new slab follow the period's last fragment read (k-step 1); what was split before that waits in the registers its float4 came in.
    auto period = [&](int stage, int slab, auto setc, auto fastc) __attribute__((always_inline)) {
        const char* base = smem + stage * STAGE;
        const char* Ahi = base; const char* Alo = base + A_PLANE;
        const char* Bhi = base + P * A_PLANE; const char* Blo = Bhi + B_PLANE;
        constexpr int NP = P == 2 ? 3 : 1, NM = NP * WTM * WTN, KS = BK / 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8_t ah[WTM], al[WTM], bh[WTN], bl[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                const int off = a_off + (4 * ks * MB + 2 * i) * 128;
                ah[i] = read8(Ahi, off, MB);
                if (P == 2) al[i] = read8(Alo, off, MB);
            }
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int off = b_off + (4 * ks * NB + 2 * j) * 128;
                bh[j] = read8(Bhi, off, NB);
                if (P == 2) bl[j] = read8(Blo, off, NB);
            }
            if (ks == KS - 1) {              // the last fragment reads are issued: the chunks split so far may go to the other stage
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (((KS - 1) * NM * NCH) / (KS * NM) > c) write_chunk(c, slab);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int pass = m / (WTM * WTN), ij = m % (WTM * WTN), i = ij / WTN, j = ij % WTN;
                // pass order as before: lo*hi, hi*lo, hi*hi (same-accumulator MFMAs stay WTM x WTN apart)
                if (P == 2 && pass == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                else if (P == 2 && pass == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                const int g = ks * NM + m;                                   // MFMAs issued before this one
                const int c0 = (g * NCH) / (KS * NM), c1 = ((g + 1) * NCH) / (KS * NM);      // chunks [c0, c1) ride behind this MFMA
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    if (c >= c0 && c < c1) {
                        split_chunk(c, slab, setc, fastc);
                        if (ks == KS - 1) write_chunk(c, slab);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using FAST = std::integral_constant<bool, true>;
    using SLOW = std::integral_constant<bool, false>;
    // branch-free pipeline over an EVEN number of slabs (an odd count gets one phantom slab of zeros): slab i in LDS stage i & 1,
    // the loads of slab i+2 in flight while slab i is multiplied and slab i+1 is split into the other stage; one barrier per slab
    load(0, S0{}, SLOW{});
    load(1, S1{}, SLOW{});
    store(0, S0{});
    __syncthreads();
    int it = 0;
    for (; it + 3 < full_slabs; it += 2) {          // full slabs only: the period touches slabs it+1 .. it+3 (splits it+1, it+2; loads it+2, it+3)
        load(it + 2, S0{}, FAST{});
        period(0, it + 1, S1{}, FAST{});
        __syncthreads();
        load(it + 3, S1{}, FAST{});
        period(1, it + 2, S0{}, FAST{});
        __syncthreads();
    }
    for (; it < nslab; it += 2) {                   // the end of the K range: clamped addresses, rows past the end zeroed
        load(it + 2, S0{}, SLOW{});
        period(0, it + 1, S1{}, SLOW{});
        __syncthreads();
        load(it + 3, S1{}, SLOW{});
        period(1, it + 2, S0{}, SLOW{});
        __syncthreads();
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
            if (p.splits > 1) p.cs_part[(long)split * p.M + m0 + i] = t;     // one row per K-split, summed by the slab reduction
            else p.colsum[m0 + i] += t;                                          // (this workgroup is the only writer of these columns)
        }
    }

    // epilogue: the MFMA ran as D = B-tile^T x A-tile -> D[row = n][col = m]; a lane holds four consecutive n per register quad
    float* slab = p.splits > 1 ? p.slab + (long)split * p.M * p.N : nullptr;
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
                float* dst = p.C + (long)row * p.ldc + col;
                if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                if (p.accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                *reinterpret_cast<float4*>(dst) = v;
            }
    }
}

int g_tn_enable = 1, g_tn_force_cfg = -1, g_tn_force_splits = 0;
int g_tn_grid_xcd = -1;          // -1: from MDVIT_TN_GRID_XCD (default 1) at the first launch

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
    // 1.5 workgroups per CU in total (tools/tn_check.py --sweep: 256 .. 512 workgroups is the optimum on every shape of the model),
    // at least 8 slabs of 32 tokens each per workgroup
    long want = g_tn_force_splits > 0 ? g_tn_force_splits : (384 + tiles - 1) / tiles;
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
    snprintf(out, cap, "gemm_tn_kernel<%d, %d, %d, %s, %s>%s", TN_BM[pl.cfg], TN_BN[pl.cfg], (d->precision == 2 && d->conv_c <= 0) ? 1 : 2,
             d->colsum_a ? "true" : "false", d->conv_c > 0 ? "true" : "false", pl.splits > 1 ? "+splitk_reduce" : "");
}

int mdvit_gemm_tn_launch(const MdvitGemmDesc* d, hipStream_t s) {
    const TnPlan pl = plan_tn(d->M, d->N, d->K, d->allow_split);
    TnArgs a;
    memset(&a, 0, sizeof(a));
    a.A = d->A; a.B = d->B; a.C = d->C; a.colsum = d->colsum_a; a.bias = d->bias;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.M = d->M; a.N = d->N; a.K = d->K;
    a.kps = pl.kps; a.splits = pl.splits; a.tiles_m = pl.tiles_m; a.tiles_n = pl.tiles_n; a.accumulate = d->accumulate;
    if (g_tn_grid_xcd < 0) { const char* e = getenv("MDVIT_TN_GRID_XCD"); g_tn_grid_xcd = (e && e[0] == '0') ? 0 : 1; }
    a.grid_xcd = g_tn_grid_xcd;
    if (d->conv_c > 0) {
        a.cv_c = d->conv_c; a.cv_h = d->conv_h; a.cv_w = d->conv_w; a.cv_ho = d->conv_ho; a.cv_wo = d->conv_wo; a.cv_s = d->conv_stride; a.cv_d = d->conv_dilation;
    }
    if (pl.splits > 1) {
        const size_t need = sizeof(float) * (size_t)pl.splits * d->M * (d->N + (d->colsum_a ? 1 : 0));
        MDVIT_CHECK_ARG(d->ws != nullptr && d->ws_bytes >= need, MDVIT_E_WORKSPACE,
                        "gemm (wgrad): split reduction needs %zu bytes of workspace (mdvit_gemm_ws_bytes), got %zu", need, (size_t)d->ws_bytes);
        a.slab = (float*)d->ws;
        a.cs_part = a.slab + (size_t)pl.splits * d->M * d->N;
    }
    const dim3 grid(pl.tiles_m * pl.tiles_n, pl.splits), block(NTH);
    const bool one = d->precision == 2;
#define MDVIT_TN_LAUNCH(BM_, BN_)                                                                                   \
    do {                                                                                                            \
        if (a.cv_c > 0) {                                                                                           \
            if (a.colsum) MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, true, true>), grid, block, 0, s, a);      \
            else MDVIT_TIMED_LAUNCH((gemm_tn_kernel<BM_, BN_, 2, false, true>), grid, block, 0, s, a);              \
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
        int rc = mdvit_gemm_splitk_reduce(a.slab, d->bias, d->C, d->ldc, d->M, d->N, pl.splits, d->accumulate, s);
        if (rc == MDVIT_OK && d->colsum_a)        // the K-splits' column-sum rows, added in slab order
            rc = mdvit_reduce_partials(a.cs_part, pl.splits, d->M, d->M, d->colsum_a, 0, nullptr, 1, s);
        return rc;
    }
    return MDVIT_OK;
}

extern "C" int mdvit_gemm_tn_config(int32_t enable, int32_t cfg, int32_t splits) {
    g_tn_enable = enable != 0;
    g_tn_force_cfg = (cfg >= 0 && cfg <= 3) ? cfg : -1;
    g_tn_force_splits = splits > 0 ? splits : 0;
    return MDVIT_OK;
}
