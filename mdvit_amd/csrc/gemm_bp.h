// Shared between the plane GEMM kernels (gemm_bp.hip: 128 / 64 tiles, lock-step slabs; gemm_ph.hip: 256-wide tiles, phase-split waves):
// the flattened launch arguments and the epilogue ids.  Library-internal.
#pragma once
#include "common.h"

struct BpArgs {
    const void* A; long lda; long a_plane; int a_f32;       // bf16 planes (lda, a_plane in elements) or fp32 [M,K]
    const uint16_t* B; long ldb; long b_plane;
    int M, N, K;
    float* C; long ldc;                                       // fp32 result (optional)
    uint16_t* Cp; long ldcp; long c_plane;                    // bf16-plane result (optional)
    float* U; long ldu_out;                                   // GELU: the pre-activation, fp32 (optional)
    const float* bias;
    int e_drop; uint32_t e_k0, e_k1, e_thresh; float e_inv_keep;
    const float* e_rowscale; int e_rows_per_scale;
    const float* residual; long ldr;
    const float* gelu_u; long ldu;
    const uint16_t* rc_a; long rc_lda; long rc_a_plane; const uint16_t* rc_b; long rc_ldb; long rc_b_plane; const float* rc_bias; int rc_k;
    int splits; int k_per_split; float* slab;
    int accumulate;
    const uint32_t* seed;
    int tiles_m, tiles_n;
};

enum { BEPI_PLAIN = 0, BEPI_GELU = 1, BEPI_DGELU = 2, BEPI_FULL = 3, BEPI_DGELU_RC = 4 };

// gemm_ph.hip: the phase-split 8-wave kernels (cfg 3: 256 x 256, cfg 4: 256 x 128, cfg 5: 128 x 256 output tile).  planes = 2 (bf16x3) / 1 (bf16);
// epi = BEPI_*; a.tiles_m / tiles_n / splits / k_per_split set by the caller for the cfg's tile.  Returns 0, or 1 when the combination is not built.
int mdvit_gemm_ph_launch(const BpArgs& a, int cfg, int planes, int epi, hipStream_t s);
bool mdvit_gemm_ph_ok(const BpArgs& a, int cfg, int planes, int epi, int kps);

// gemm_pm.hip: the phase-split 8-wave kernels on a 128-row tile (cfg 6: 128 x 160, cfg 7: 128 x 128) for the mid-size products; fp32 A, two weight planes, one K range.
int mdvit_gemm_pm_launch(const BpArgs& a, int cfg, int epi, hipStream_t s);
bool mdvit_gemm_pm_ok(const BpArgs& a, int cfg, int planes, int epi);
extern "C" int mdvit_gemm_pm_prefers(int32_t M, int32_t N, int32_t K, int32_t planes, int32_t a_f32);
extern "C" int mdvit_gemm_pm_splits(int32_t M, int32_t N, int32_t K, int32_t planes, int32_t a_f32);
