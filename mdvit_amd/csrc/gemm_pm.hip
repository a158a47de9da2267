// Phase-split plane GEMM on a 128-row tile: C[M,N] = A[M,K] B[N,K]^T for the MID-SIZE products of the C = 320 / 512 blocks (proj, fc2, the qkv / fc1 data gradients
// and their forward layers at 16-128 images: mdvit.py:267,307, mpvit.py:71-78) -- the shapes where 256 x 256 tiles (gemm_ph.hip) leave half the chip empty and the
// 64 / 128 lock-step tiles (gemm.hip, gemm_bp.hip) reach ~22 % of the bf16x3 matrix roof (tools/gemm_shapes_time.py, profiles/r05_gemm_shapes_alone.txt).
//
//   * Output tile 128 x BN, BN = 32 (NX + NY): 160 (NX = 3, NY = 2) for N = 320 / 960 / 1280 -- 16384 x 320 is exactly 256 workgroups, one per CU -- or 128 (2, 2).
//     EIGHT waves: wave w and wave w + 4 share SIMD w and the tile's row block w (32 rows); waves 0-3 ("X") own the first NX 32 x 32 column blocks of it, waves 4-7
//     ("Y") the other NY.  The two waves of a SIMD do unequal work, the four SIMDs equal work.
//   * Same operands and arithmetic as gemm_bp.hip / gemm_ph.hip (fp32 A split while it is staged, the weight as its per-step bf16 planes; per accumulator and
//     k step of 16: lo*hi, hi*lo, hi*hi on v_mfma_f32_32x32x16_bf16 in ascending k): results are BIT-IDENTICAL to their tiles, so the planner may pick by shape.
//   * A K tile is 32 k.  Ring of FOUR LDS stages, each A hi | A lo (128 rows x 64 B) | B hi | B lo (BN rows x 64 B).  B travels L2 -> LDS by global_load_lds
//     (1 KiB = 16 rows per wave-instruction), A HBM -> registers (three sets in flight) -> planes -> ds_write_b128.
//   * Y runs ONE BARRIER behind X: a K tile is a load phase L (fragment reads of tile u, the LDS-DMA issue of B(u+3), the register loads of A(u+4), counted vmcnt) and a
//     multiply phase M (MFMAs); while X multiplies, Y loads, and the other way round.  The conversion of an older tile's A registers (A(u+2): 8 v_mov + the split + 2
//     ds_write_b128) sits where its half has room: in X's LOAD phase (X has 18 MFMAs per tile) and in Y's MULTIPLY phase (12 MFMAs).  Schedule (interval k between
//     barriers k, k+1):
//         X:  L(0) M(0) L(1) M(1) ...
//         Y:       L(0) M(0) L(1) ...
//     RAW: data is waited for / written at least one barrier before the first phase that reads it (A(u+2): written in intervals 2u (X) / 2u+2 (Y), read from 2u+4;
//     B(u+1): waited for at the end of L(u), intervals 2u / 2u+1, read from 2u+2).  WAR: B(u+3) goes to stage (u+3) & 3, which held tile u-1 -- read in L(u-1), intervals
//     2u-2 / 2u-1, every wave's reads COMPLETE (lgkmcnt(0)) before the barrier that ends its load phase; A(u+2) goes to the stage of tile u-2.  Tiles past the end are
//     loaded from the last tile's address into stages nobody reads again, so every phase issues the same number of vector-memory
//     operations and the vmcnt immediates are constants.
//
// Measured (tools/gemm_pm_check.py, profiles/r05_gemm_pm_check.txt): 16384 x 320 x 1280 in 44-46 us = ~300 TF/s useful against 200 on gemm.hip's 64 x 64 tiles -- a K tile
// takes ~2600 cycles against 960 of MFMA.  Two other main loops were built on the same ring and measure THE SAME: this tile's vector-memory issue dealt between the MFMA
// groups instead of behind the first one, and all eight waves in step with ONE barrier per K tile and the fragment registers as the pipeline
// (profiles/r05_gemm_pm_schedules.txt: 45.5 against 44.0 us).  Counters (profiles/r05_gemm_pm_pmc_stalls.txt): matrix pipes 36 % busy, a wave waits 44 % of its life, 4.4 VALU
// instructions per MFMA, no LDS bank conflicts.  NOT the weight's traffic either: every workgroup streams the whole weight from L2 in 64-byte row pieces (half of every
// 128-byte line), but a probe build with k-tile-major planes -- every 1 KiB piece one contiguous KiB -- runs the one-round shapes no faster (16384 x 320 x 1280 54.5 against
// 49.4 us, x 960 38.0 / 39.0; only the eight-round 131072 x 320 x 1280 gains, 357 against 411: tools/probe/gemm_pm_tiled_probe.py, profiles/r05_gemm_pm_tiled_probe.txt).
// Shader-clock stamps (tools/probe/gemm_pm_phases.py on the -DMDVIT_PM_STAMPS build, profiles/r05_gemm_pm_phases.txt) say what is: per K tile a SIMD issues 30 MFMAs
// (960 cycles) + 28 ds_read_b128 + 10 vector-memory operations + ~76 VALU instructions from its two waves, and the tile takes ~2080 cycles wherever the non-MFMA
// instructions stand -- inside the multiply phases (first build) or in the load phases (now): the two waves of a SIMD slow each other's issue down by what they overlap, as
// profiles/r05_mfma_valu_overlap.txt found for VALU.  At 3 MFMAs per four fragment reads (bf16x3 on 32 x 32 blocks) this tile shape has too few MFMAs per other instruction.
#include "common.h"
#include "gemm_bp.h"
#include <type_traits>

typedef float pm_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 pm_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pm_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pm_u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PM_THREADS = 512;
constexpr int PM_AP = 128 * 64;          // bytes of one A plane of a stage

#define PM_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define PM_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("s_barrier" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// Variant build -DMDVIT_PM_STAMPS (tools/build_variant.py; tools/probe/gemm_pm_phases.py): shader-clock stamps of lane 0 of waves 0 (X) and 4 (Y) of workgroup 0 around the
// parts of a K tile, written to the buffer handed in as BpArgs.slab (unused otherwise: the kernel has one K range): [wave half][tile][8]
#ifdef MDVIT_PM_STAMPS
#define PM_STAMP(t_, k_) do { if (stamps && (t_) < 64) stamps[(t_) * 8 + (k_)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define PM_STAMP(t_, k_) do { } while (0)
#endif

__device__ __forceinline__ int pm_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int NX, int NY, int EPI>
__global__ __launch_bounds__(PM_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2), amdgpu_num_vgpr(224))) void gemm_pm_kernel(BpArgs p) {
    constexpr int BN = 32 * (NX + NY);
    constexpr int BP = BN * 64;                              // bytes of one B plane of a stage
    constexpr int STAGE = 2 * PM_AP + 2 * BP;
    constexpr int NPIECE = 2 * BN / 16, PPW = (NPIECE + 7) / 8;      // 1 KiB LDS-DMA pieces of a stage's B planes; per wave
    constexpr int NV = PPW + 2;                              // vector-memory operations a wave issues per multiply phase
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5, wr = wave & 3;
    uint32_t s0 = 0, s1 = 0;
    if (p.seed) { s0 = p.seed[0]; s1 = p.seed[1]; }
    const uint32_t ek0 = p.e_k0 ^ s0, ek1 = p.e_k1 + s1;
    const int tile = pm_xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
    const int m0 = tm * 128, n0 = tn * BN;
    // a K split (grid.y; BEPI_PLAIN only, gemm_bp.hip's planner): this workgroup multiplies k = k0 .. k0 + 32 nt - 1 and leaves the raw sums in its slab
    // [split][M][N]; gemm_splitk_reduce_kernel adds the slabs in split order (+ bias, accumulate)
    const bool split = EPI == BEPI_PLAIN && p.splits > 1;
    const int k0 = split ? (int)blockIdx.y * p.k_per_split : 0;
    const int nt = (split ? min(p.k_per_split, p.K - k0) : p.K) / 32;

    // ---- B: wave w brings pieces w, w + 8 (, w + 16): piece pc = plane pc / (BN/16), rows 16 (pc % (BN/16)) ..+15; lane i lands at row (i >> 2), physical 16-byte chunk
    // i & 3 and fetches logical chunk (i & 3) ^ ((row >> 2) & 3) (the swizzle lives on the SOURCE address: the LDS side of global_load_lds is lane-linear).  A piece index
    // past the last one repeats the wave's previous piece (same bytes to the same place).
    static_assert(PPW <= 3, "three pieces per wave at most");
    const uint16_t* gb[3]; int ob[3];          // (fixed extents, and a local copy at the builtin below: with an element of a template-sized array as the builtin's
    //                                             argument hipcc's HOST pass silently emits no stub for the kernel -- the library then fails to load)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int pc = wave + 8 * i;
        if (pc >= NPIECE) pc -= 8;
        const int pl = pc / (BN / 16), rg = pc % (BN / 16);
        const int row = rg * 16 + (lane >> 2), chunk = (lane & 3) ^ ((row >> 2) & 3);
        gb[i] = p.B + (long)pl * p.b_plane + (long)min(n0 + row, p.N - 1) * p.ldb + chunk * 8 + k0;
        ob[i] = 2 * PM_AP + pl * BP + rg * 1024;
    }
    auto issue_b1 = [&](int t, int i) __attribute__((always_inline)) {
        const uint16_t* g = gb[i] + (long)min(t, nt - 1) * 32;
        char* dst = smem + (t & 3) * STAGE + ob[i];
        __builtin_amdgcn_global_load_lds(g, dst, 16, 0, 0);
    };
    auto issue_b = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_b1(t, i);
    };
    // ---- A (fp32): thread (row = tid / 4, c = tid % 4) owns k = 8 c .. 8 c + 7 of its row of a tile: two 16-byte loads into one of three register sets
    const int qrow = tid >> 2, qc = tid & 3;
    const float* ga = reinterpret_cast<const float*>(p.A) + (long)min(m0 + qrow, p.M - 1) * p.lda + qc * 8 + k0;
    // The three in-flight A sets live in FIXED registers v[232:239], v[240:247], v[248:255], above the range the compiler may allocate (amdgpu_num_vgpr(224) on the
    // kernel): as compiler-visible asm outputs they were COPIED while still in flight -- hipcc placed v_mov_b64 of the prologue's A(2) / A(3) registers in front of the loop
    // (register assignment at a control-flow merge), and a copy of a register whose load has not landed copies garbage (the first build of this kernel: results changed
    // from run to run).  A set is read back (v_mov into compiler registers) only behind its counted s_waitcnt.  S is a constant after inlining.
    auto load_a = [&](int t, int S) __attribute__((always_inline)) {
        const float* g = ga + (long)min(t, nt - 1) * 32;
        if (S == 0) asm volatile("global_load_dwordx4 v[232:235], %0, off\n\tglobal_load_dwordx4 v[236:239], %0, off offset:16" ::"v"(g)
                                 : "memory", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239");
        else if (S == 1) asm volatile("global_load_dwordx4 v[240:243], %0, off\n\tglobal_load_dwordx4 v[244:247], %0, off offset:16" ::"v"(g)
                                      : "memory", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
        else asm volatile("global_load_dwordx4 v[248:251], %0, off\n\tglobal_load_dwordx4 v[252:255], %0, off offset:16" ::"v"(g)
                          : "memory", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
    };
    auto read_set = [&](int S, float (&f)[8]) __attribute__((always_inline)) {
#define PM_RD8(b0, b1, b2, b3, b4, b5, b6, b7)                                                                                                                        \
    asm volatile("v_mov_b32 %0, v" #b0 "\n\tv_mov_b32 %1, v" #b1 "\n\tv_mov_b32 %2, v" #b2 "\n\tv_mov_b32 %3, v" #b3 "\n\tv_mov_b32 %4, v" #b4 "\n\tv_mov_b32 %5, v" #b5       \
                 "\n\tv_mov_b32 %6, v" #b6 "\n\tv_mov_b32 %7, v" #b7                                                                                                   \
                 : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]), "=v"(f[6]), "=v"(f[7])::"memory")
        if (S == 0) PM_RD8(232, 233, 234, 235, 236, 237, 238, 239);
        else if (S == 1) PM_RD8(240, 241, 242, 243, 244, 245, 246, 247);
        else PM_RD8(248, 249, 250, 251, 252, 253, 254, 255);
#undef PM_RD8
    };
    const int woff = qrow * 64 + ((qc ^ ((qrow >> 2) & 3)) << 4);
    auto lds_store16 = [&](uint32_t addr, const uint4 v4) __attribute__((always_inline)) {
        const pm_u32x4 v = {v4.x, v4.y, v4.z, v4.w};
        asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
    };
    auto write_a = [&](int t, int S) __attribute__((always_inline)) {
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem) + (t & 3) * STAGE + woff;
        float f[8];
        read_set(S, f);
        uint2 h0, l0, h1, l1;
        mdvit_split_bf16x3(make_float4(f[0], f[1], f[2], f[3]), h0, l0);
        mdvit_split_bf16x3(make_float4(f[4], f[5], f[6], f[7]), h1, l1);
        lds_store16(dst, make_uint4(h0.x, h0.y, h1.x, h1.y));
        lds_store16(dst + PM_AP, make_uint4(l0.x, l0.y, l1.x, l1.y));
    };

    // ---- fragments: lane (l31, lhi) of a 32-row block reads row l31, logical chunk 2 ks + lhi; ks = 1 flips bit 5 of the byte offset
    const int swz = (l31 >> 2) & 3;
    const int fr = l31 * 64 + ((lhi ^ swz) << 4);

    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>; using C2 = std::integral_constant<int, 2>;

    // prologue: tiles 0 and 1 of A converted, B(0) landed; then, in the order the multiply phases M(-2), M(-1) would have issued them: B(1) A(2) B(2) A(3)
    issue_b(0);
    load_a(0, 0);
    load_a(1, 1);
    PM_WAIT_VM(0);
    __builtin_amdgcn_sched_barrier(0);
    write_a(0, 0);
    write_a(1, 1);
    issue_b(1);
    load_a(2, 2);
    issue_b(2);
    load_a(3, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PM_BAR();

    auto run = [&](auto grpc) __attribute__((always_inline)) {
        constexpr int G = decltype(grpc)::value;
        constexpr int NC = G == 0 ? NX : NY, CB0 = G == 0 ? 0 : NX;
        pm_f32x16 acc[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        pm_bf16x8 af[2][2], bf[NC][2][2];                     // [plane][k step]
#ifdef MDVIT_PM_STAMPS
        long long* stamps = (p.slab && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) ? reinterpret_cast<long long*>(p.slab) + (wave >> 2) * 64 * 8 : nullptr;
#endif
        // one K tile: L(t), barrier, M(t), barrier.  U = t % 3: M(t) loads A(t + 4) into set (U + 1) % 3 and converts A(t + 2) from set (U + 2) % 3
        auto tile_body = [&](int t, auto uc) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value;
            const char* st = smem + (t & 3) * STAGE;
            PM_STAMP(t, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    af[h][ks] = __builtin_bit_cast(pm_bf16x8, *reinterpret_cast<const uint4*>(st + h * PM_AP + wr * 2048 + (ks ? (fr ^ 32) : fr)));
#pragma unroll
                    for (int j = 0; j < NC; ++j)
                        bf[j][h][ks] = __builtin_bit_cast(pm_bf16x8, *reinterpret_cast<const uint4*>(st + 2 * PM_AP + h * BP + (CB0 + j) * 2048 + (ks ? (fr ^ 32) : fr)));
                }
            // The tile's vector-memory issue and the conversion of an older tile's A registers are NOT in the multiply phase of the half that has the most MFMAs: stamped
            // (tools/probe/gemm_pm_phases.py, profiles/r05_gemm_pm_phases.txt), a multiply phase that carried both took its MFMA time PLUS ~450 cycles (3 LDS-DMA issues,
            // 2 loads, 8 v_mov + the 24-instruction split + 2 ds_write: the matrix pipe idles behind a wave's non-MFMA issue), 1032 / 838 cycles for X / Y, while the other
            // half's load phase (~390) waited at the barrier -- 2078 cycles per K tile.  Now both halves issue in their LOAD phase; X (18 MFMAs per tile) also converts
            // there, Y (12 MFMAs) converts in its multiply phase: the two intervals of a tile hold max(X.M, Y.L) and max(Y.M, X.L) with the extras on the short sides.
            issue_b(t + 3);
            load_a(t + 4, (U + 1) % 3);
            PM_STAMP(t, 1);
            if constexpr (G == 0) {
                PM_WAIT_VM(2 * NV);                           // A(t + 2) is in its registers (younger: B(t + 2), A(t + 3), B(t + 3), A(t + 4)) -- and B(t + 1), older, has landed
                __builtin_amdgcn_sched_barrier(0);
                write_a(t + 2, (U + 2) % 3);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                PM_WAIT_VM(2 + 2 * NV);                       // B(t + 1) has landed (younger: A(t + 2), B(t + 2), A(t + 3), B(t + 3), A(t + 4))
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this tile's fragment reads are DONE before the barrier: X refills stage (t) & 3 ... (t + 4) from its next load phase on
            }
            PM_STAMP(t, 2);
            PM_BAR();
            PM_STAMP(t, 3);
            __builtin_amdgcn_s_setprio(1);
            auto group = [&](int g) __attribute__((always_inline)) {
                const int ks = g / 3, pr = g % 3;
#pragma unroll
                for (int j = 0; j < NC; ++j)
                    acc[j] = pr == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][1][ks], af[0][ks], acc[j], 0, 0, 0)
                           : (pr == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0][ks], af[1][ks], acc[j], 0, 0, 0)
                                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][0][ks], af[0][ks], acc[j], 0, 0, 0));
                __builtin_amdgcn_sched_barrier(0);
            };
            group(0);
            group(1);
            group(2);
            group(3);
            PM_STAMP(t, 4);
            if constexpr (G == 1) {
                PM_WAIT_VM(2 * NV);                           // A(t + 2) is in its registers
                __builtin_amdgcn_sched_barrier(0);
                PM_STAMP(t, 5);
                write_a(t + 2, (U + 2) % 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            group(4);
            group(5);
            PM_STAMP(t, 6);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PM_BAR();
            PM_STAMP(t, 7);
        };
        if (G == 1) PM_BAR();                                 // the second half of the workgroup runs one barrier behind the first
        for (int t = 0; t < nt; t += 3) {
            tile_body(t, C0{});
            if (t + 1 < nt) tile_body(t + 1, C1{});
            if (t + 2 < nt) tile_body(t + 2, C2{});
        }
        if (G == 0) PM_BAR();
        PM_WAIT_VM(0);                                        // the tail tiles' loads (registers of this wave, LDS nobody reads) before the epilogue's own counting

        // ---- epilogue (gemm_bp.hip's arithmetic on this kernel's block map): the MFMA ran as B^T x A -> D[row = n][col = m] per 32 x 32 block: for each register
        // quad q a lane holds FOUR CONSECUTIVE output columns n = 8 q + 4 (lane >> 5) + (r & 3) of output row m = lane & 31.  Loads are unconditional (clamped address,
        // the value dropped by a select): a load under a branch makes every later store wait for vmcnt(0).
        constexpr bool HAS_IN = EPI == BEPI_PLAIN || EPI == BEPI_DGELU || EPI == BEPI_FULL;
        const bool use_in = EPI == BEPI_PLAIN ? (p.accumulate != 0 && !split) : (EPI == BEPI_DGELU ? true : p.residual != nullptr);
        const float* in_p = EPI == BEPI_PLAIN ? p.C : (EPI == BEPI_DGELU ? p.gelu_u : p.residual);
        const long in_ld = EPI == BEPI_PLAIN ? p.ldc : (EPI == BEPI_DGELU ? p.ldu : p.ldr);
        if (!use_in || in_p == nullptr) in_p = reinterpret_cast<const float*>(p.B);
        const bool use_bias = p.bias != nullptr && !split;
        float* const c_out = split ? p.slab + (long)blockIdx.y * p.M * p.N : p.C;
        const long c_ld = split ? (long)p.N : p.ldc;
        uint16_t* const cp_out = split ? nullptr : p.Cp;
        const float* bias_p = use_bias ? p.bias : reinterpret_cast<const float*>(p.B);
        const int row = m0 + wr * 32 + l31;
        const int rowc = min(row, p.M - 1);
        const bool row_ok = row < p.M;
        float rsc = 1.f;
        if constexpr (EPI == BEPI_FULL) {
            const bool use_rs = p.e_rowscale != nullptr;
            const float* rs_p = use_rs ? p.e_rowscale : reinterpret_cast<const float*>(p.B);
            const float v = rs_p[use_rs ? rowc / p.e_rows_per_scale : 0];
            rsc = use_rs ? v : 1.f;
        }
        float4 inq[2][4];
        auto load_in = [&](int j) __attribute__((always_inline)) {
            if constexpr (HAS_IN) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = min(n0 + (CB0 + j) * 32 + 8 * q + 4 * lhi, p.N - 4);
                    inq[j & 1][q] = *reinterpret_cast<const float4*>(in_p + (use_in ? (long)rowc * in_ld + col : 0L));
                }
            }
        };
        load_in(0);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            if (j + 1 < NC) load_in(j + 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = n0 + (CB0 + j) * 32 + 8 * q + 4 * lhi;
                const bool ok = row_ok && col < p.N;
                float4 v = make_float4(acc[j][4 * q + 0], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
                {
                    const float4 b4 = *reinterpret_cast<const float4*>(bias_p + (use_bias ? min(col, p.N - 4) : 0));
                    if (use_bias) { v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                }
                const uint32_t didx = (uint32_t)((long)row * p.N + col);
                float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (HAS_IN) { o4 = inq[j & 1][q]; if (EPI != BEPI_DGELU && !use_in) o4 = make_float4(0.f, 0.f, 0.f, 0.f); }
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
                    v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;       // (a multiply and an add, never fused: gemm_bp.hip / gemm_ph.hip / gemm_body.inc do the same arithmetic, bit for bit)
                    v.x += o4.x; v.y += o4.y; v.z += o4.z; v.w += o4.w;
                }
                if (c_out && ok) *reinterpret_cast<float4*>(c_out + (long)row * c_ld + col) = v;
                if (cp_out && ok) {
                    uint2 hi, lo;
                    mdvit_split_bf16x3(v, hi, lo);
                    uint16_t* d = cp_out + (long)row * p.ldcp + col;
                    *reinterpret_cast<uint2*>(d) = hi;
                    *reinterpret_cast<uint2*>(d + p.c_plane) = lo;
                }
            }
        }
    };
    if (wave < 4) run(C0{});
    else run(C1{});
}

#ifdef MDVIT_PM_STAMPS
static long long* g_pm_stamps = nullptr;
extern "C" void mdvit_pm_debug_buffer(long long* buf) { g_pm_stamps = buf; }
#endif

template <int NX, int NY>
int launch_pm(const BpArgs& a0, int epi, hipStream_t s) {
    BpArgs a = a0;
#ifdef MDVIT_PM_STAMPS
    if (a.splits > 1) return 1;          // (the stamps travel in the slab pointer: the probe build has one K range)
    a.slab = reinterpret_cast<float*>(g_pm_stamps);
#endif
    constexpr int BN = 32 * (NX + NY), LDS = 4 * (2 * PM_AP + 2 * BN * 64);
    dim3 grid(a.tiles_m * a.tiles_n, a.splits > 1 ? a.splits : 1), block(PM_THREADS);
#define PM_LAUNCH(EPI_)                                                                                                                         \
    do {                                                                                                                                        \
        static bool attr[64];                                                                                                                   \
        int dev = 0;                                                                                                                            \
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;                                                                  \
        if (!attr[dev]) {                                                                                                                       \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pm_kernel<NX, NY, EPI_>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return 2; \
            attr[dev] = true;                                                                                                                   \
        }                                                                                                                                       \
        MDVIT_TIMED_LAUNCH((gemm_pm_kernel<NX, NY, EPI_>), grid, block, LDS, s, a);                                                            \
    } while (0)
    switch (epi) {
        case BEPI_PLAIN: PM_LAUNCH(BEPI_PLAIN); break;
        case BEPI_GELU: PM_LAUNCH(BEPI_GELU); break;
        case BEPI_DGELU: PM_LAUNCH(BEPI_DGELU); break;
        case BEPI_FULL: PM_LAUNCH(BEPI_FULL); break;
        default: return 1;
    }
#undef PM_LAUNCH
    return 0;
}

}  // namespace

// cfg 6: 128 x 160, cfg 7: 128 x 128.  Built for what the step hands in: fp32 A, two weight planes (bf16x3), one K range, K % 32 == 0.
bool mdvit_gemm_pm_ok(const BpArgs& a, int cfg, int planes, int epi) {
    if (!((cfg == 6 || cfg == 7) && planes == 2 && a.a_f32 && epi != BEPI_DGELU_RC && a.K % 32 == 0 && a.K >= 32 && a.N % 4 == 0)) return false;
    // K splits (round 5, last): plain epilogue, no plane output, every split a whole number of 32-wide K tiles (four at least: the ring's prologue), the slab handed in
    return a.splits == 1 || (epi == BEPI_PLAIN && !a.Cp && a.slab && a.k_per_split % 32 == 0 && a.k_per_split >= 128 && a.K - (a.splits - 1) * a.k_per_split >= 128);
}

int mdvit_gemm_pm_launch(const BpArgs& a, int cfg, int epi, hipStream_t s) {
    if (cfg == 6) return launch_pm<3, 2>(a, epi, s);
    if (cfg == 7) return launch_pm<2, 2>(a, epi, s);
    return 1;
}

// Does the 128-row phase-split kernel take this NT product?  mode -1: never, 0: by the rule, 1: whenever legal (tools/gemm_ph_check.py).  Returns the cfg (6 / 7) or 0.
int g_pm_mode = 0;
extern "C" int mdvit_gemm_pm_config(int32_t mode) {
    g_pm_mode = mode;
    return MDVIT_OK;
}
extern "C" int mdvit_gemm_pm_prefers(int32_t M, int32_t N, int32_t K, int32_t planes, int32_t a_f32) {
    if (g_pm_mode < 0 || planes != 2 || !a_f32 || M <= 0 || N <= 0 || K < 64 || K % 32 != 0 || N % 4 != 0) return 0;
    const int cfg = N % 160 == 0 ? 6 : (N % 128 == 0 ? 7 : 0);
    if (cfg == 0) return 0;
    if (g_pm_mode > 0) return cfg;
    // One workgroup per CU: what decides is how much of the chip's round(s) of tiles is real work, and whether the K loop is long enough to pay for a prologue and an
    // epilogue that nothing overlaps.  Measured against gemm.hip (tools/gemm_pm_check.py, profiles/r05_gemm_pm_check.txt): 16384 x 320 x 320 / 960 / 1280 (one round)
    // 18 / 36 / 44 us against 23 / 55 / 67, 8192 x 512 x 512 / 2048 20 / 56 against 27 / 80, 32768 x 320 x 1280 (two rounds) 88 against 110; with K = 320 over
    // several rounds it loses (16384 x 1280 x 320 65 against 54, 32768 x 960 x 320 94 against 85); 131072 x 320 x 1280 (eight rounds) ties.
    const long tiles = (long)cdiv(M, 128) * (N / (cfg == 6 ? 160 : 128)), rounds = (tiles + 255) / 256;
    const double eff = (double)tiles / ((double)rounds * 256.0);
    if (eff < 0.7) return 0;          // (a PLAIN product with a long K may still come here as K splits: mdvit_gemm_pm_splits)
    return ((K >= 640 && rounds <= 4) || (K >= 256 && rounds == 1)) ? cfg : 0;
}
// The K splits of a PLAIN product whose tiles leave half of the chip idle and whose K is long: 4096 x 512 x 2048 (the fc1 data gradient of stage 3 at 16 images, 128 tiles of
// 128 x 128) as two K ranges of 1024 fills the 256 CUs; the slabs go through gemm_splitk_reduce_kernel in split order.  Measured alone (tools/gemm_pm_check.py,
// profiles/r05_gemm_pm_splits.txt): 42.9 us with the reduction against 50.0 on the 64 x 64 tile of gemm.hip it ran before (and 50.8 unsplit).  Ranges shorter than 1024
// LOSE -- 4096 x 512 x 1536 as 2 x 768: 37.6 against 33.8; 2048 x 512 x 2048 as 4 x 512: 33.0 against 31.2; 4096 x 320 x 1280 as 2 x 640: 32.9 against 28.4 (the reduction
// launch and a prologue / epilogue per range that nothing overlaps) -- so the rule asks for ranges >= 1024 and >= 192 workgroups.  In the step: +0.3 % (profiles/r05_ab_pm_split.txt,
// within the boxes' noise).  1: no split (or not this kernel's product).  MDVIT_PM_SPLIT=0: never.
extern "C" int mdvit_gemm_pm_splits(int32_t M, int32_t N, int32_t K, int32_t planes, int32_t a_f32) {
    static const bool on = [] { const char* e = getenv("MDVIT_PM_SPLIT"); return !(e && e[0] == '0'); }();
    if (!on || g_pm_mode != 0 || planes != 2 || !a_f32 || M <= 0 || N <= 0 || K < 1024 || K % 32 != 0 || N % 4 != 0) return 1;
    const int cfg = N % 160 == 0 ? 6 : (N % 128 == 0 ? 7 : 0);
    if (cfg == 0) return 1;
    const long tiles = (long)cdiv(M, 128) * (N / (cfg == 6 ? 160 : 128));
    if (tiles > 128 || tiles < 32) return 1;
    int best = 1;
    for (int s = 2; s <= 4; ++s) {
        const int kps = cdiv(cdiv(K, s), 32) * 32;
        if (kps < 1024 || K - (cdiv(K, kps) - 1) * kps < 128) continue;
        const long wgs = tiles * cdiv(K, kps);
        if (wgs >= 192 && wgs <= 256) best = cdiv(K, kps);
    }
    return best;
}
