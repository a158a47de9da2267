// Phase-split plane GEMM: C[M,N] = A[M,K] B[N,K]^T on a 256 x 256 output tile with EIGHT waves (512 threads, one workgroup per CU),
// for the MFMA-bound layers (C >= 320: qkv / proj / fc1 / fc2 and their data gradients, mdvit.py:267,307, mpvit.py:71-78; the bridge and
// linear_fuse products).  Same operands, arithmetic and epilogues as gemm_bp.hip (bf16 planes, hi*lo + lo*hi + hi*hi per k step in ascending
// k, fp32 accumulate: results are bit-identical to its tiles); what differs is the main loop:
//
//   * A K tile (32 k for bf16x3, 64 k for the single-plane mode) of the 256 x 256 problem is FOUR "units" of 128 rows x 2 x 64 B
//     (A0 A1 B0 B1; the two 64-byte halves are the hi / lo planes, or the two 32-wide k halves of the single-plane mode).  Units travel
//     HBM / L2 -> LDS by global_load_lds (1 KiB per wave-instruction, two per wave and unit) into a two-stage ring, ONE unit per phase.
//   * A wave owns rows {64 wr .. +64} of BOTH A units and columns {32 wc .. +32} of BOTH B units (wr = wave / 4, wc = wave % 4), so the
//     four quadrants of its 128 x 64 accumulator are (A0,B0) (A0,B1) (A1,B1) (A1,B0) for every wave: a phase = one quadrant x one K tile =
//     12 MFMAs (32x32x16) behind at most 12 ds_read_b128; fragments are re-used across neighbouring quadrants, so a unit is read from LDS
//     in exactly one phase per tile (A0, B0: phase 0; B1: phase 1; A1: phase 2) and its slot is re-staged two phases later.
//   * Waves 4-7 run ONE BARRIER behind waves 0-3 (they share the SIMDs pairwise): while one half multiplies, the other half issues its
//     LDS reads and its global_load_lds and waits at the barrier -- the matrix pipe of every SIMD always has a wave in its MFMA phase.
//   * Loads are never drained: each phase ends its load part with a COUNTED s_waitcnt vmcnt(8) (four units stay in flight across the
//     barriers; vector-memory operations retire in order) and raw s_barrier; a unit is waited for one phase before it is read
//     (the guide's rule for LDS-DMA data with two staggered wave groups), and re-staged no earlier than two phases after its last read.
//
// The structure is the guide's 256^2 8-phase template (cdna_hip_programming.md section 5) re-derived for two-plane operands.
#include "common.h"
#include "gemm_bp.h"

typedef float ph_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ph_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ph_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ph_u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PH_THREADS = 512;
constexpr int UP = 128 * 64;            // bytes of one plane (or k half) of a unit: 128 rows x 64 B
constexpr int UNIT = 2 * UP;
constexpr int STAGE = 4 * UNIT;         // A0 A1 B0 B1
enum { U_A0 = 0, U_A1 = 1, U_B0 = 2, U_B1 = 3 };

#define PH_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define PH_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("s_barrier" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

__device__ __forceinline__ int ph_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// AF32: A is fp32 [M, K] in HBM and is split while it is staged: the A units go HBM -> registers (global_load_dwordx4 in inline asm: loads hipcc must not
// count, or it would drain the LDS-DMA queue at their first use) -> hi / lo planes -> ds_write_b128, the conversion and the LDS write placed behind the first
// MFMAs of a phase (VALU and LDS issue ride in the matrix pipe's shadow).  vmcnt is counted by hand over BOTH kinds of loads (they retire in issue order).
//
// Slot schedule (phase s = 4 t + q of K tile t; "M(s)" = the MFMA part of phase s, "L(s)" its load part):
//   M(4t+0): A1(t) registers -> LDS [AF32];  issue A1(t+1)         L(4t+0) reads A0(t), B0(t)
//   M(4t+1): issue B0(t+2)                                         L(4t+1) reads B1(t)
//   M(4t+2): A0(t+1) registers -> LDS [AF32];  issue A0(t+2)       L(4t+2) reads A1(t)
//   M(4t+3): issue B1(t+2)                                         L(4t+3) reads nothing
// L(s) ends with s_waitcnt vmcnt(what M(s-3) .. M(s-1) issued): everything issued in M(s-4) or earlier has landed -- which covers every unit L(s+1) reads and
// every register M(s) converts.  RAW (guide, two staggered wave groups): data is waited for / written one phase before the phase that reads it.  WAR: a slot is
// re-staged no earlier than the MFMA part of the phase after its last read (all waves have passed the lgkmcnt(0) behind that read by then).
//
// (A two-tile-deep prefetch of the fp32 A rows -- two register sets per unit -- was built and does not fit: 128 accumulator + 48 fragment + 32 in-flight registers
//  plus the conversion's temporaries exceed 256, and what hipcc spills are the in-flight registers themselves.  The streamed operand is what holds this loop:
//  tools/probe/gemm_ph_cached_probe.py, with the A rows cached the same launch runs 25 % faster, with B cached 8 %.)
template <int P, bool AF32, int EPI>
__global__ __launch_bounds__(PH_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_ph_kernel(BpArgs p) {
    constexpr int KT = P == 2 ? 32 : 64;
    constexpr int LA = (AF32 && P == 1) ? 4 : 2;            // vector-memory operations of an A slot (B slots: 2)
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5, wr = wave >> 2, wc = wave & 3;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t ek0 = p.e_k0 ^ s0, ek1 = p.e_k1 + s1;
    const int tile = ph_xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;           // the column tiles of a row panel are neighbours on one XCD (the other order -- row panels
    //                                                                    of a column tile adjacent -- measured slower on 14 of 17 shapes: tools/probe/gemm_ph_order_probe.py history)
    const int m0 = tm * 256, n0 = tn * 256;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nt = (kend - kbeg) / KT;

    // ---- staging by global_load_lds: wave w copies rows 16 w .. 16 w + 15 of both halves of a unit; lane i lands at row 16 w + (i >> 2), physical 16-byte chunk
    // i & 3 and fetches logical chunk (i & 3) ^ ((row >> 2) & 3) of that row (the swizzle lives on the SOURCE address: the LDS side of global_load_lds is lane-linear)
    const int prow = 16 * wave + (lane >> 2), pchunk = (lane & 3) ^ ((lane >> 4) & 3);
    const uint16_t* gp[4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ra = min(m0 + u * 128 + prow, p.M - 1), rb = min(n0 + u * 128 + prow, p.N - 1);   // rows past the edge: any valid row (the epilogue discards them)
        gp[u] = AF32 ? nullptr : reinterpret_cast<const uint16_t*>(p.A) + (long)ra * p.lda + kbeg + pchunk * 8;
        gp[2 + u] = p.B + (long)rb * p.ldb + kbeg + pchunk * 8;
    }
    const long a_h1 = P == 2 ? p.a_plane : 32, b_h1 = P == 2 ? p.b_plane : 32;         // second half of a unit: the lo plane, or k + 32
    auto issue = [&](int u, int t) __attribute__((always_inline)) {
        const uint16_t* g = gp[u] + (long)t * KT;
        char* dst = smem + (t & 1) * STAGE + u * UNIT + wave * 1024;
        __builtin_amdgcn_global_load_lds(g, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g + (u < 2 ? a_h1 : b_h1), dst + UP, 16, 0, 0);
    };
    // ---- staging of fp32 A through registers: thread (row = tid / 4, c = tid % 4) owns k = 8 c .. 8 c + 7 of its row of a unit (and of k + 32 .. for the second
    // k half of the single-plane mode)
    const int qrow = tid >> 2, qc = tid & 3;
    const float* ga[2];
    ph_f32x4 rga[2][1][LA];           // in-flight A0 / A1 rows
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ra = min(m0 + u * 128 + qrow, p.M - 1);
        ga[u] = AF32 ? reinterpret_cast<const float*>(p.A) + (long)ra * p.lda + kbeg + qc * 8 : nullptr;
    }
    const int woff = qrow * 64 + ((qc ^ ((qrow >> 2) & 3)) << 4);
    auto load_regs = [&](int u, int t, int set = 0) __attribute__((always_inline)) {
        const float* g = ga[u] + (long)t * KT;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rga[u][set][0]) : "v"(g) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(rga[u][set][1]) : "v"(g) : "memory");
        if constexpr (LA == 4) {
            asm volatile("global_load_dwordx4 %0, %1, off offset:128" : "=v"(rga[u][set][2]) : "v"(g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:144" : "=v"(rga[u][set][3]) : "v"(g) : "memory");
        }
    };
    // (the LDS stores are inline asm too: hipcc orders a ds_write it can see behind EVERY outstanding global_load_lds -- s_waitcnt vmcnt(0) -- as a possible
    //  write-after-write on LDS; their completion is waited for by hand, lgkmcnt(0) in front of the phase's closing barrier)
    auto lds_store16 = [&](uint32_t addr, const uint4 v4) __attribute__((always_inline)) {
        const ph_u32x4 v = {v4.x, v4.y, v4.z, v4.w};
        asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
    };
    auto write_regs = [&](int u, int t, int set = 0) __attribute__((always_inline)) {
        const ph_f32x4 (&rg)[LA] = rga[u][set];
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem) + (t & 1) * STAGE + u * UNIT + woff;
        if constexpr (P == 2) {
            uint2 h0, l0, h1, l1;
            mdvit_split_bf16x3(make_float4(rg[0][0], rg[0][1], rg[0][2], rg[0][3]), h0, l0);
            mdvit_split_bf16x3(make_float4(rg[1][0], rg[1][1], rg[1][2], rg[1][3]), h1, l1);
            lds_store16(dst, make_uint4(h0.x, h0.y, h1.x, h1.y));
            lds_store16(dst + UP, make_uint4(l0.x, l0.y, l1.x, l1.y));
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint2 h0, l0, h1, l1;
                mdvit_split_bf16x3(make_float4(rg[2 * h][0], rg[2 * h][1], rg[2 * h][2], rg[2 * h][3]), h0, l0);
                mdvit_split_bf16x3(make_float4(rg[2 * h + 1][0], rg[2 * h + 1][1], rg[2 * h + 1][2], rg[2 * h + 1][3]), h1, l1);
                lds_store16(dst + h * UP, make_uint4(h0.x, h0.y, h1.x, h1.y));
            }
        }
    };
    auto issue_a = [&](int u, int t) __attribute__((always_inline)) {
        if constexpr (AF32) load_regs(u, t); else issue(u, t);
    };

    // ---- fragments: lane (l31, lhi) of a 32-row block reads row l31, logical chunk 2 ks + lhi; chunk ^ swizzle: ks = 1 flips bit 5 of the byte offset
    const int swz = (l31 >> 2) & 3;
    const int fa = (wr * 64 + l31) * 64 + ((lhi ^ swz) << 4);
    const int fb = (wc * 32 + l31) * 64 + ((lhi ^ swz) << 4);
    ph_bf16x8 af[2][2][2];            // [block of the unit][half][k step]
    ph_bf16x8 bf0[2][2], bf1[2][2];   // [half][k step] of B0 / B1
    auto load_a = [&](int unit, const char* st) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    af[b][h][ks] = __builtin_bit_cast(ph_bf16x8, *reinterpret_cast<const uint4*>(st + unit * UNIT + h * UP + b * 2048 + (ks ? (fa ^ 32) : fa)));
    };
    auto load_b = [&](ph_bf16x8 (&bf)[2][2], int unit, const char* st) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bf[h][ks] = __builtin_bit_cast(ph_bf16x8, *reinterpret_cast<const uint4*>(st + unit * UNIT + h * UP + (ks ? (fb ^ 32) : fb)));
    };

    ph_f32x16 acc[4][2];              // [2 * A unit + block][B unit]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one quadrant x one K tile; the MFMA runs as B^T x A (a lane then owns four consecutive output columns); per accumulator and k step: lo*hi, hi*lo, hi*hi
    auto mma = [&](int i0, int j, const ph_bf16x8 (&bf)[2][2], auto&& mid) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (P == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[i0 + b][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[1][ks], af[b][0][ks], acc[i0 + b][j], 0, 0, 0);
                if (ks == 0) mid();
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[i0 + b][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0][ks], af[b][1][ks], acc[i0 + b][j], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[i0 + b][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0][ks], af[b][0][ks], acc[i0 + b][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[i0 + b][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[h][ks], af[b][h][ks], acc[i0 + b][j], 0, 0, 0);
                    if (h == 0 && ks == 0) mid();
                }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // One K tile = four phases.  V1 / V2: tile t + 1 / t + 2 exists (their slots issue); W0 .. W3: vmcnt at the end of the phases' load parts.
    auto tile_body = [&](int t, auto v1, auto v2, auto w0, auto w1, auto w2, auto w3) __attribute__((always_inline)) {
        constexpr bool V1 = decltype(v1)::value, V2 = decltype(v2)::value;
        const char* st = smem + (t & 1) * STAGE;
        load_b(bf0, U_B0, st);
        load_a(U_A0, st);
        PH_WAIT_VM(decltype(w0)::value);
        PH_BAR();
        mma(0, 0, bf0, [&]() __attribute__((always_inline)) {
            if constexpr (AF32) write_regs(1, t);
            if constexpr (V1) issue_a(U_A1, t + 1);
        });
        if constexpr (AF32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PH_BAR();
        load_b(bf1, U_B1, st);
        PH_WAIT_VM(decltype(w1)::value);
        PH_BAR();
        mma(0, 1, bf1, [&]() __attribute__((always_inline)) { if constexpr (V2) issue(U_B0, t + 2); });
        PH_BAR();
        load_a(U_A1, st);
        PH_WAIT_VM(decltype(w2)::value);
        PH_BAR();
        mma(2, 1, bf1, [&]() __attribute__((always_inline)) {
            if constexpr (AF32 && V1) write_regs(0, t + 1);
            if constexpr (V2) issue_a(U_A0, t + 2);
        });
        if constexpr (AF32 && V1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PH_BAR();
        PH_WAIT_VM(decltype(w3)::value);
        PH_BAR();
        mma(2, 0, bf0, [&]() __attribute__((always_inline)) { if constexpr (V2) issue(U_B1, t + 2); });
        PH_BAR();
    };
    using T_ = std::true_type; using F_ = std::false_type;
#define PH_I(n) std::integral_constant<int, (n)>{}

    // prologue: what M(-7) .. M(-1) would have issued, in their order: B0(0) A0(0) B1(0) A1(0) B0(1) A0(1) B1(1)
    issue(U_B0, 0);
    issue_a(U_A0, 0);
    issue(U_B1, 0);
    issue_a(U_A1, 0);
    issue(U_B0, 1);
    if constexpr (AF32) {
        PH_WAIT_VM(2 + LA + 2);                 // A0(0) is in its registers
        __builtin_amdgcn_sched_barrier(0);
        write_regs(0, 0);
    }
    issue_a(U_A0, 1);
    issue(U_B1, 1);
    PH_WAIT_VM(2 + LA + 2);
    if constexpr (AF32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PH_BAR();
    if (wave >= 4) PH_BAR();                    // the second half of the workgroup runs one barrier behind the first
    for (int t = 0; t < nt - 2; ++t) tile_body(t, T_{}, T_{}, PH_I(2 + LA + 2), PH_I(LA + 2 + LA), PH_I(2 + LA + 2), PH_I(LA + 2 + LA));
    tile_body(nt - 2, T_{}, F_{}, PH_I(2 + LA + 2), PH_I(LA + 2 + LA), PH_I(2 + LA), PH_I(LA));
    tile_body(nt - 1, F_{}, F_{}, PH_I(0), PH_I(0), PH_I(0), PH_I(0));
    if (wave < 4) PH_BAR();

    // ---- epilogue (gemm_bp.hip's arithmetic on this kernel's block map): D[row = n][col = m] per 32x32 block -- for each register quad q a lane holds FOUR
    // CONSECUTIVE output columns n = 8 q + 4 (lane >> 5) + (r & 3) of output row m = lane & 31.
    // Every LOAD is unconditional (clamped address, the value dropped by a select): hipcc puts s_waitcnt vmcnt(0) behind a branch that holds a load, and vmcnt
    // counts the stores too -- a bias / residual / gelu_u quad fetched under `if` made every store wait for all the stores before it.  The operand rows of
    // row block i + 1 are requested before block i is stored (the wait for them leaves block i's stores in flight).
    const bool split = (EPI == BEPI_PLAIN) && p.splits > 1;
    float* slab = split ? p.slab + (long)blockIdx.y * p.M * p.N : nullptr;
    constexpr bool HAS_IN = EPI == BEPI_PLAIN || EPI == BEPI_DGELU || EPI == BEPI_FULL;       // an [M, N] fp32 operand read in the epilogue (old C | gelu_u | residual)
    const bool use_in = EPI == BEPI_PLAIN ? (p.accumulate != 0 && !split) : (EPI == BEPI_DGELU ? true : p.residual != nullptr);
    const float* in_p = EPI == BEPI_PLAIN ? p.C : (EPI == BEPI_DGELU ? p.gelu_u : p.residual);
    const long in_ld = EPI == BEPI_PLAIN ? p.ldc : (EPI == BEPI_DGELU ? p.ldu : p.ldr);
    if (!use_in || in_p == nullptr) in_p = reinterpret_cast<const float*>(p.B);               // any readable address: the value is dropped
    const bool use_bias = p.bias != nullptr && !split;
    const float* bias_p = use_bias ? p.bias : reinterpret_cast<const float*>(p.B);
    float4 bq[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = min(n0 + j * 128 + wc * 32 + 8 * q + 4 * lhi, p.N - 4);
            const float4 b4 = *reinterpret_cast<const float4*>(bias_p + (use_bias ? col : 0));
            bq[j][q] = use_bias ? b4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    float rsc[4];
    if constexpr (EPI == BEPI_FULL) {
        const bool use_rs = p.e_rowscale != nullptr;
        const float* rs_p = use_rs ? p.e_rowscale : reinterpret_cast<const float*>(p.B);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(m0 + (i >> 1) * 128 + wr * 64 + (i & 1) * 32 + l31, p.M - 1);
            const float v = rs_p[use_rs ? row / p.e_rows_per_scale : 0];
            rsc[i] = use_rs ? v : 1.f;
        }
    }
    constexpr int IND = 3;            // blocks of the epilogue operand in flight (one workgroup per CU: nothing else hides their latency)
    float4 inq[IND][4];
    auto load_in = [&](int ij) __attribute__((always_inline)) {
        if constexpr (HAS_IN) {
            const int i = ij >> 1, j = ij & 1;
            const int row = min(m0 + (i >> 1) * 128 + wr * 64 + (i & 1) * 32 + l31, p.M - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = min(n0 + j * 128 + wc * 32 + 8 * q + 4 * lhi, p.N - 4);
                inq[ij % IND][q] = *reinterpret_cast<const float4*>(in_p + (use_in ? (long)row * in_ld + col : 0L));
            }
        }
    };
#pragma unroll
    for (int ij = 0; ij < IND - 1; ++ij) load_in(ij);
#pragma unroll
    for (int ij = 0; ij < 8; ++ij) {
        const int i = ij >> 1, j = ij & 1;
        if (ij + IND - 1 < 8) load_in(ij + IND - 1);
        const int row = m0 + (i >> 1) * 128 + wr * 64 + (i & 1) * 32 + l31;
        const bool row_ok = row < p.M;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + j * 128 + wc * 32 + 8 * q + 4 * lhi;
                const bool ok = row_ok && col < p.N;
                float4 v = make_float4(acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                if (split) { if (ok) *reinterpret_cast<float4*>(slab + (long)row * p.N + col) = v; continue; }
                { const float4 b4 = bq[j][q]; v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                const uint32_t didx = (uint32_t)((long)row * p.N + col);
                float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (HAS_IN) { o4 = inq[ij % IND][q]; if (EPI != BEPI_DGELU && !use_in) o4 = make_float4(0.f, 0.f, 0.f, 0.f); }
                if (EPI == BEPI_PLAIN) { v.x += o4.x; v.y += o4.y; v.z += o4.z; v.w += o4.w; }
                if (EPI == BEPI_GELU) {
                    if (p.U && ok) *reinterpret_cast<float4*>(p.U + (long)row * p.ldu_out + col) = v;
                    v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
                }
                if (EPI == BEPI_DGELU) { v.x *= gelu_grad_f(o4.x); v.y *= gelu_grad_f(o4.y); v.z *= gelu_grad_f(o4.z); v.w *= gelu_grad_f(o4.w); }
                if (EPI != BEPI_PLAIN && p.e_drop) {
                    const float4 ds = mdvit_drop_scale4(ek0, ek1, didx, p.e_thresh, p.e_inv_keep);
                    v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
                }
                if (EPI == BEPI_FULL) {
#pragma clang fp contract(off)
                    const float rs = rsc[i];       // (a multiply and an add, never fused: gemm_bp.hip / gemm_body.inc / lin_rc_kernel do the same arithmetic, bit for bit)
                    v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
                    v.x += o4.x; v.y += o4.y; v.z += o4.z; v.w += o4.w;
                }
                if (p.C && ok) *reinterpret_cast<float4*>(p.C + (long)row * p.ldc + col) = v;
                if (p.Cp && ok) {
                    uint2 hi, lo;
                    mdvit_split_bf16x3(v, hi, lo);
                    uint16_t* d = p.Cp + (long)row * p.ldcp + col;
                    *reinterpret_cast<uint2*>(d) = hi;
                    if (P == 2) *reinterpret_cast<uint2*>(d + p.c_plane) = lo;
                }
            }
        }
    }
}

template <int P, bool AF32>
int launch_ph(const BpArgs& a, int epi, hipStream_t s) {
    dim3 grid(a.tiles_m * a.tiles_n, a.splits), block(PH_THREADS);
#define PH_LAUNCH(EPI_) MDVIT_TIMED_LAUNCH((gemm_ph_kernel<P, AF32, EPI_>), grid, block, 0, s, a)
    switch (epi) {
        case BEPI_PLAIN: PH_LAUNCH(BEPI_PLAIN); break;
        case BEPI_GELU: PH_LAUNCH(BEPI_GELU); break;
        case BEPI_DGELU: PH_LAUNCH(BEPI_DGELU); break;
        case BEPI_FULL: PH_LAUNCH(BEPI_FULL); break;
        default: return 1;
    }
#undef PH_LAUNCH
    return 0;
}

}  // namespace

bool mdvit_gemm_ph_ok(const BpArgs& a, int cfg, int planes, int epi, int kps) {
    if (cfg != 3 || epi == BEPI_DGELU_RC) return false;
    const int kt = planes == 2 ? 32 : 64;
    if (a.K % kt != 0 || kps % kt != 0 || kps < 2 * kt) return false;
    const int last = a.K - (a.splits - 1) * kps;        // the last split's extent
    return last >= 2 * kt && last % kt == 0;
}

int mdvit_gemm_ph_launch(const BpArgs& a, int cfg, int planes, int epi, hipStream_t s) {
    if (cfg != 3) return 1;
    if (a.a_f32) return planes == 2 ? launch_ph<2, true>(a, epi, s) : launch_ph<1, true>(a, epi, s);
    return planes == 2 ? launch_ph<2, false>(a, epi, s) : launch_ph<1, false>(a, epi, s);
}

// Does the 256-wide kernel take this NT product?  Its workgroups own a whole CU: what decides is how much of the chip's round(s) of 256 x 256 tiles is real
// output.  Measured against the 64 / 128 tiles at 2-5 workgroups per CU (tools/gemm_ph_check.py, fp32 A, one MI355X): it wins from ~0.72 (8192 x 1536 x 512:
// 43.7 against 46.5 us at 0.75; 32768 x 960 x 320: 69.9 / 74.5 at 0.94; 8192 x 2048 x 512: 51.5 / 61.0 at 1.0) and loses below (16384 x 1280 x 320:
// 63.9 / 49.1 at 0.625; 4096 x 2048 x 512: 38.4 / 32.0 at 0.5).  mode -1: never, 0: by that rule, 1: whenever the shape is legal.
// epi_reads != 0: the epilogue READS an [M, N] operand (gelu_u of the fc2 data gradient, the residual of proj / fc2, the old C of an accumulating launch).  With
// one workgroup per CU nothing hides those loads' latency (the 64 / 128 tiles overlap them with a neighbour's main loop): 32768 x 1280 x 320 with the gelu'
// epilogue runs 174-177 us here against 153-158 us on gemm.hip although the plain product ties (0.83 of three rounds); 8192 x 2048 x 512 (one full round) wins
// 61-62 against 76-77.  Such launches are taken from 0.9.
int g_ph_mode = 0;
extern "C" int mdvit_gemm_ph_prefers_epi(int32_t M, int32_t N, int32_t K, int32_t planes, int32_t epi_reads) {
    const int kt = planes == 2 ? 32 : 64;
    if (g_ph_mode < 0 || M <= 0 || N <= 0 || K < 2 * kt || K % kt != 0 || N % 4 != 0) return 0;
    if (g_ph_mode > 0) return 1;
    const long tiles = (long)cdiv(M, 256) * cdiv(N, 256), rounds = (tiles + 255) / 256;
    const double eff = (double)M * N / ((double)rounds * 256.0 * 65536.0);
    return eff >= (epi_reads ? 0.9 : 0.72) ? 1 : 0;
}
extern "C" int mdvit_gemm_ph_prefers(int32_t M, int32_t N, int32_t K, int32_t planes) { return mdvit_gemm_ph_prefers_epi(M, N, K, planes, 0); }

extern "C" int mdvit_gemm_ph_config(int32_t mode) {
    g_ph_mode = mode;
    return MDVIT_OK;
}
