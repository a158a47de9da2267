// One C-ABI entry per SerialBlock_adapt pass (mdvit.py:316-361): the forward / backward of a whole block is ENQUEUED from C -- its ~15 / ~35
// kernels, the fork of the weight-gradient work onto the side stream, the carving of one caller-owned buffer into the block's saved and
// temporary tensors -- instead of ~25 Python-level operator calls with an autograd node, a handful of torch.empty and a ctypes
// round trip each.  Round 2 measured the host at 30-38 ms per step against 31 ms of main-stream kernel time (profiles/r02k_*): the step
// was bound by the enqueue rate, and the 16 blocks are half of its launches.
// The kernels and their order are exactly those of the operator-level path (mdvit_amd/ops.py: dwconv3x3 + layer_norm + linear + factor_att +
// linear + layer_norm + mlp_residual and their backward), so both paths give bit-identical results (tests/test_gpu_block.py).
#include <stdlib.h>
#include "common.h"

namespace {

struct Arena {
    char* base; size_t off, cap; bool dry;
    float* take(size_t floats) {
        const size_t bytes = (floats * sizeof(float) + 255) & ~(size_t)255;
        float* p = dry ? nullptr : reinterpret_cast<float*>(base + off);
        off += bytes;
        return p;
    }
    void* take_bytes(size_t bytes) { return take((bytes + 3) / 4); }
};

enum { MLP_RC = 0, MLP_RECOMP = 1, MLP_STORED = 2 };

inline int mlp_mode(const MdvitBlockDesc& d) {
    if (d.precision == 1 && d.C == 64 && d.hidden % 256 == 0 && d.hidden <= 4096 && d.fc1_p && d.fc2_p && (long)d.B * d.H * d.W * d.hidden < (1L << 32)) return MLP_RC;
    if (d.precision >= 1 && d.C <= 128 && d.C % 32 == 0 && d.hidden % 4 == 0) return MLP_RECOMP;
    return MLP_STORED;
}
// MLP_RECOMP at C = 128 with the weight planes given: forward and backward data path as ONE kernel each (mlp_rc.hip, 16-token waves); h and du still
// reach HBM once each, for the two weight-gradient GEMMs
inline bool mlp_rc16(const MdvitBlockDesc& d) {
    return mlp_mode(d) == MLP_RECOMP && d.precision == 1 && d.C == 128 && d.hidden % 32 == 0 && d.hidden <= 4096 && d.fc1_p && d.fc2_p && d.fc1_b && d.fc2_b &&
           (long)d.B * d.H * d.W * d.hidden < (1L << 32);
}

// "mixed" mode: h and du of the C = 128 MLP as bf16 (they exist only on the 16-token kernels' path)
inline bool hbf16(const MdvitBlockDesc& d) { return d.store_bf16 && mlp_rc16(d); }

struct Saved {        // the block's saved-for-backward tensors inside the caller's `save` buffer
    float *x1, *mean1, *rstd1, *cur1, *qkv, *a, *att, *U, *kmax, *ksum, *Mmat, *x2, *mean2, *rstd2, *cur2, *h, *u, *lse;
};

inline bool sdpa_kind(const MdvitBlockDesc& d) { return d.attn_kind == 1; }

void layout_saved(const MdvitBlockDesc& d, Arena& A, Saved& s) {
    const long T = (long)d.B * d.H * d.W, C = d.C, Ch = d.C / d.heads;
    const int mode = mlp_mode(d);
    s.lse = nullptr;
    s.x1 = sdpa_kind(d) ? nullptr : A.take(T * C);          // (Block_adapt: no ConvPosEnc -- x1 IS the block's input, which the caller keeps)
    s.mean1 = A.take(T); s.rstd1 = A.take(T); s.cur1 = A.take(T * C);
    s.qkv = A.take(T * 3 * C);
    s.a = d.label ? A.take((long)d.B * C) : nullptr;
    if (d.label && d.a_pre) s.a = const_cast<float*>(d.a_pre);          // computed ahead for every adapter of the network (mdvit_da_fwd_many): the slot above stays unused
    s.att = A.take(T * C);
    if (sdpa_kind(d)) { s.U = s.kmax = s.ksum = s.Mmat = nullptr; s.lse = A.take((long)d.B * d.heads * d.H * d.W); }
    else {
    s.U = A.take(T * C);
    s.kmax = A.take((long)d.B * C); s.ksum = A.take((long)d.B * C); s.Mmat = A.take((long)d.B * C * Ch);
    }
    s.x2 = A.take(T * C); s.mean2 = A.take(T); s.rstd2 = A.take(T); s.cur2 = A.take(T * C);
    s.h = mode != MLP_RC ? A.take(hbf16(d) ? (T * d.hidden + 1) / 2 : T * d.hidden) : nullptr;
    s.u = mode == MLP_STORED ? A.take(T * d.hidden) : nullptr;
}

void gemm_init(MdvitGemmDesc& g, const MdvitBlockDesc& d) {
    memset(&g, 0, sizeof(g));
    g.precision = d.precision >= 1 ? 1 : 0;
    g.drop_seed = nullptr;
}

// the C = 64 MLP backward as ONE kernel (mdvit_mlp_rc_bwd): 0 never, 1 always (default: >= neutral in the three-stream step in both arithmetic modes, see mdvit_amd/ops.py),
// 2 when the call has no weight-gradient stream (tuning hook: mdvit_block_config)
static int g_blk_mlp_bwd = 1;
extern "C" int mdvit_block_config(int32_t mlp_bwd_fused) {
    if (mlp_bwd_fused < 0 || mlp_bwd_fused > 2) return mdvit_set_error(MDVIT_E_SHAPE, "block_config: mlp_bwd_fused in 0..2");
    g_blk_mlp_bwd = mlp_bwd_fused;
    return MDVIT_OK;
}

#define BLK_RUN(call)                         \
    do {                                      \
        if (!A.dry) {                         \
            const int rc__ = (call);          \
            if (rc__ != MDVIT_OK) return rc__; \
        }                                     \
    } while (0)

// An NT product of the block on the 256-wide plane kernel (gemm_ph.hip through mdvit_gemm_planes: fp32 activations split while staged, the weight as its
// per-step bf16 planes [2][N][K]) -- when the weight's planes were handed in and mdvit_gemm_ph_prefers takes the shape.  Same arithmetic, same results.
// ... or on the 128-row phase-split tile of the mid-size products (gemm_pm.hip, round 5; mdvit_gemm_planes picks between the two by the same rules)
bool ph_takes(const MdvitBlockDesc& d, const void* planes, int M, int N, int K, int epi_reads) {
    return d.precision == 1 && planes != nullptr && (mdvit_gemm_ph_prefers_epi(M, N, K, 2, epi_reads) != 0 || mdvit_gemm_pm_prefers(M, N, K, 2, 1) != 0);
}
int epi_reads_of(const MdvitGemmDesc& g) { return g.gelu_u != nullptr || g.residual != nullptr || g.accumulate != 0; }
// ... or, a plain product with a long K and few tiles (the stage-3 data gradients at 16 images), on that tile over 2-4 K ranges (mdvit_gemm_pm_splits)
bool pm_split_takes(const MdvitBlockDesc& d, const MdvitGemmDesc& g, const void* planes, int M, int N, int K) {
    return d.precision == 1 && planes != nullptr && g.allow_split && g.epi == MDVIT_EPI_NONE && !epi_reads_of(g) && !(g.e_drop_p > 0.f) && !g.e_rowscale && !g.rc_a &&
           mdvit_gemm_pm_splits(M, N, K, 2, 1) > 1;
}
int gemm_planes_nt(Arena& A, const MdvitGemmDesc& g, const void* planes, hipStream_t s, bool k_splits = false) {
    MdvitPlaneGemmDesc pd;
    memset(&pd, 0, sizeof(pd));
    pd.A = g.A; pd.lda = g.lda; pd.a_f32 = 1;
    pd.B = planes; pd.ldb = g.K; pd.b_plane = (int64_t)g.N * g.K;
    pd.planes = 2; pd.M = g.M; pd.N = g.N; pd.K = g.K;
    if (g.epi == MDVIT_EPI_GELU_DUAL && g.C2) { pd.U = g.C; pd.ldu_out = g.ldc; pd.C = g.C2; pd.ldc = g.ldc; }
    else { pd.C = g.C; pd.ldc = g.ldc; }
    pd.bias = g.bias; pd.epi = g.epi;
    pd.e_drop_p = g.e_drop_p; pd.e_key0 = g.e_key0; pd.e_key1 = g.e_key1; pd.e_rowscale = g.e_rowscale; pd.e_rows_per_scale = g.e_rows_per_scale;
    pd.residual = g.residual; pd.ldr = g.ldr; pd.gelu_u = g.gelu_u; pd.ldu = g.ldu;
    pd.drop_seed = g.drop_seed;
    pd.allow_split = k_splits ? 1 : 0;          // (only where pm_split_takes routed the product here: the planner's other routes keep their one K range)
    const size_t need = pd.allow_split ? mdvit_gemm_planes_ws_bytes(&pd) : 0;          // the slabs of the K splits
    pd.ws = need ? A.take_bytes(need) : nullptr; pd.ws_bytes = need;
    BLK_RUN(mdvit_gemm_planes(&pd, s));
    return MDVIT_OK;
}

// C = A W^T (+ epilogue): the forward layers
int gemm_fwd(Arena& A, MdvitGemmDesc& g, hipStream_t s, const MdvitBlockDesc* d = nullptr, const void* planes = nullptr) {
    if (d && ph_takes(*d, planes, g.M, g.N, g.K, epi_reads_of(g))) return gemm_planes_nt(A, g, planes, s);
    g.trans_a = 0; g.trans_b = 1; g.allow_split = 1;
    const size_t need = mdvit_gemm_ws_bytes(&g);
    g.ws = need ? A.take_bytes(need) : nullptr; g.ws_bytes = need;
    BLK_RUN(mdvit_gemm_f32(&g, s));
    return MDVIT_OK;
}

// dx[M,K] = g[M,N] W[N,K]: NT against the cached W^T (bf16x3) or NN (fp32)  -- ops._dgrad
int gemm_dgrad(Arena& A, const MdvitBlockDesc& d, MdvitGemmDesc& g, const float* gy, const float* W, const float* Wt, float* dx, int M, int K, int N, hipStream_t s,
               const void* planes_t = nullptr) {
    g.A = gy; g.C = dx; g.M = M; g.N = K; g.K = N; g.lda = N; g.ldc = K;
    if (ph_takes(d, planes_t, M, K, N, epi_reads_of(g)) && !g.rc_a) return gemm_planes_nt(A, g, planes_t, s);
    if (pm_split_takes(d, g, planes_t, M, K, N)) return gemm_planes_nt(A, g, planes_t, s, true);
    if (d.precision >= 1) { g.B = Wt; g.ldb = N; g.trans_a = 0; g.trans_b = 1; g.precision = 1; }
    else { g.B = W; g.ldb = K; g.trans_a = 0; g.trans_b = 0; g.precision = 0; }
    const size_t need = g.allow_split ? mdvit_gemm_ws_bytes(&g) : 0;
    g.ws = need ? A.take_bytes(need) : nullptr; g.ws_bytes = need;
    BLK_RUN(mdvit_gemm_f32(&g, s));
    return MDVIT_OK;
}

// dW[N,K] (+)= gy[M,N]^T x[M,K], db[N] += colsum(gy)  -- the TN launch of ops._Linear.backward
int gemm_wgrad(Arena& A, const MdvitBlockDesc& d, const float* gy, const float* x, float* dW, float* db, int M, int N, int K, int accumulate, hipStream_t s,
               int gy_bf16 = 0, int x_bf16 = 0) {
    MdvitGemmDesc g;
    gemm_init(g, d);
    g.a_bf16 = gy_bf16; g.b_bf16 = x_bf16;
    g.A = gy; g.B = x; g.C = dW; g.M = N; g.N = K; g.K = M; g.lda = N; g.ldb = K; g.ldc = K;
    g.trans_a = 1; g.trans_b = 0; g.allow_split = 1; g.accumulate = accumulate; g.colsum_a = db;
    const size_t need = mdvit_gemm_ws_bytes(&g);
    g.ws = need ? A.take_bytes(need) : nullptr; g.ws_bytes = need;
    BLK_RUN(mdvit_gemm_f32(&g, s));
    return MDVIT_OK;
}

// everything the side stream enqueues from here on runs after what the main stream holds now
int fork_side(const MdvitBlockDesc& d, const MdvitBlockStreams& st, hipStream_t& side) {
    side = (hipStream_t)st.side;
    if (side == nullptr || side == (hipStream_t)st.main) { side = (hipStream_t)st.main; return MDVIT_OK; }
    if (st.n_events <= 0 || st.events == nullptr) return mdvit_set_error(MDVIT_E_SHAPE, "block: a side stream needs events");
    const int ei = *st.next_event;                 // wrapped explicitly: the counter only ever moved up, and a signed overflow indexed out of bounds
    *st.next_event = (ei + 1) % st.n_events;
    hipEvent_t ev = (hipEvent_t)st.events[(ei % st.n_events + st.n_events) % st.n_events];
    hipError_t e = hipEventRecord(ev, (hipStream_t)st.main);
    if (e == hipSuccess) e = hipStreamWaitEvent(side, ev, 0);
    if (e != hipSuccess) return mdvit_set_error(MDVIT_E_HIP, "block: side-stream fork failed: %s", hipGetErrorString(e));
    return MDVIT_OK;
}

int block_fwd(const MdvitBlockDesc& d, const float* x, float* y, Arena& SV, Arena& A, hipStream_t s) {
    const int B = d.B, H = d.H, W = d.W, C = d.C, Hd = d.hidden;
    const long T = (long)B * H * W;
    const int M = (int)T, N_tok = H * W;
    const int mode = mlp_mode(d);
    Saved sv;
    layout_saved(d, SV, sv);
    // x1 = x + dwconv3x3(x) + bias            (ConvPosEnc, mpvit.py:239-248)        [Block_adapt: x1 = x]
    if (sdpa_kind(d)) sv.x1 = const_cast<float*>(x);
    else BLK_RUN(mdvit_dwconv3x3_fwd(x, d.cpe_w, d.cpe_b, sv.x1, B, H, W, C, 1, 1, s));
    // cur1 = LN1(x1);  qkv = cur1 Wqkv^T + b                    (mdvit.py:286-288)
    const bool lin_rc = d.precision == 1 && (C == 64 || C == 128) && d.qkv_p && d.proj_p && M >= 1024;       // the streaming short-K Linear (mlp_rc.hip)
    static const bool ln_prologue = [] { const char* e = getenv("MDVIT_LN_PROLOGUE"); return !(e && e[0] == '0'); }();
    if (lin_rc && ln_prologue) {
        // LN1 in the qkv kernel's prologue: the rows are normalised in the registers they are multiplied from; cur1 is still written (the qkv
        // weight-gradient GEMM reads it), x1 is read once instead of twice and one launch is gone
        BLK_RUN(mdvit_linear_rc_ln(sv.x1, d.n1_g, d.n1_b, d.ln_groups, d.eps, sv.mean1, sv.rstd1, sv.cur1, d.qkv_p, 3L * C * C, d.qkv_b, sv.qkv, 3 * C, M, 3 * C, C, s));
    } else if (lin_rc) {
        BLK_RUN(mdvit_layernorm_fwd(sv.x1, d.n1_g, d.n1_b, sv.cur1, sv.mean1, sv.rstd1, M, C, d.ln_groups, d.eps, s));
        BLK_RUN(mdvit_linear_rc(sv.cur1, C, d.qkv_p, 3L * C * C, d.qkv_b, sv.qkv, 3 * C, M, 3 * C, C, 0.f, 0, 0, nullptr, 1, nullptr, 0, nullptr, s));
    } else {
        BLK_RUN(mdvit_layernorm_fwd(sv.x1, d.n1_g, d.n1_b, sv.cur1, sv.mean1, sv.rstd1, M, C, d.ln_groups, d.eps, s));
        MdvitGemmDesc g;
        gemm_init(g, d);
        g.A = sv.cur1; g.B = d.qkv_w; g.C = sv.qkv; g.M = M; g.N = 3 * C; g.K = C; g.lda = C; g.ldb = C; g.ldc = 3 * C; g.bias = d.qkv_b;
        const int rc = gemm_fwd(A, g, s, &d, d.qkv_p);
        if (rc != MDVIT_OK) return rc;
    }
    // a = softmax_heads(MLP(one_hot));  att = a * (scale * q (softmax_tokens(k)^T v) + q * crpe(v))      (mdvit.py:293-304)
    if (d.label && !d.a_pre) BLK_RUN(mdvit_da_fwd(d.label, d.da_w1, d.da_b1, d.da_w2, d.da_b2, sv.a, B, d.D, d.da_hidden, C, d.heads, s));
    if (sdpa_kind(d)) {
        // att = a * softmax(q k^T / sqrt(64)) v on the fp32 matrix cores, the row log-sum-exp kept for the backward      (vision_transformer.py:148-163)
        BLK_RUN(mdvit_sdpa_mfma_fwd(sv.qkv, sv.a, sv.att, sv.lse, B, N_tok, C, d.heads, s));
    } else {
        const size_t fab = mdvit_factoratt_ws_bytes(B, N_tok, C, d.heads);
        void* faws = A.take_bytes(fab);
        BLK_RUN(mdvit_factoratt_fwd(sv.qkv, d.w3, d.b3, d.w5, d.b5, d.w7, d.b7, sv.a, sv.att, sv.U, sv.kmax, sv.ksum, sv.Mmat, faws, fab, B, H, W, C, d.heads,
                                    d.s3, d.s5, d.s7, s));
    }
    // x2 = x1 + droppath(drop(att Wproj^T + b))      (mdvit.py:310-311,353)
    if (lin_rc) {
        BLK_RUN(mdvit_linear_rc(sv.att, C, d.proj_p, (long)C * C, d.proj_b, sv.x2, C, M, C, C, d.drop_p, d.key_proj[0], d.key_proj[1], d.rowscale1, N_tok, sv.x1, C,
                                d.drop_p > 0.f ? d.drop_seed : nullptr, s));
    } else {
        MdvitGemmDesc g;
        gemm_init(g, d);
        g.A = sv.att; g.B = d.proj_w; g.C = sv.x2; g.M = M; g.N = C; g.K = C; g.lda = C; g.ldb = C; g.ldc = C; g.bias = d.proj_b;
        g.e_drop_p = d.drop_p; g.e_key0 = d.key_proj[0]; g.e_key1 = d.key_proj[1]; g.e_rowscale = d.rowscale1; g.e_rows_per_scale = N_tok;
        g.residual = sv.x1; g.ldr = C; g.drop_seed = d.drop_p > 0.f ? d.drop_seed : nullptr;
        const int rc = gemm_fwd(A, g, s, &d, d.proj_p);
        if (rc != MDVIT_OK) return rc;
    }
    // cur2 = LN2(x2);  y = x2 + droppath(drop(fc2(drop(gelu(fc1(cur2))))))                           (mpvit.py:71-78, mdvit.py:356-360)
    const uint32_t* seed = d.drop_p > 0.f ? d.drop_seed : nullptr;
    if (ln_prologue && (mode == MLP_RC || mlp_rc16(d)) && d.hidden % 64 == 0) {
        // LN2 in the MLP kernel's prologue: the rows are normalised in the registers they are multiplied from (cur2 is still written: the backward kernels read it)
        if (hbf16(d))
            BLK_RUN(mdvit_mlp_rc_fwd_ln_hbf16(sv.x2, d.n2_g, d.n2_b, d.ln_groups, d.eps, sv.mean2, sv.rstd2, sv.cur2, d.fc1_p, d.fc1_b, d.fc2_p, d.fc2_b, d.rowscale2, N_tok,
                                              sv.h, y, M, C, Hd, d.drop_p, d.key_fc1[0], d.key_fc1[1], d.key_fc2[0], d.key_fc2[1], seed, s));
        else
            BLK_RUN(mdvit_mlp_rc_fwd_ln(sv.x2, d.n2_g, d.n2_b, d.ln_groups, d.eps, sv.mean2, sv.rstd2, sv.cur2, d.fc1_p, d.fc1_b, d.fc2_p, d.fc2_b, d.rowscale2, N_tok,
                                        mode == MLP_RC ? nullptr : sv.h, y, M, C, Hd, d.drop_p, d.key_fc1[0], d.key_fc1[1], d.key_fc2[0], d.key_fc2[1], seed, s));
        return MDVIT_OK;
    }
    BLK_RUN(mdvit_layernorm_fwd(sv.x2, d.n2_g, d.n2_b, sv.cur2, sv.mean2, sv.rstd2, M, C, d.ln_groups, d.eps, s));
    if (mode == MLP_RC) {
        BLK_RUN(mdvit_mlp_rc_fwd(sv.cur2, d.fc1_p, d.fc1_b, d.fc2_p, d.fc2_b, sv.x2, d.rowscale2, N_tok, y, M, C, Hd, d.drop_p, d.key_fc1[0], d.key_fc1[1],
                                 d.key_fc2[0], d.key_fc2[1], seed, s));
    } else if (mlp_rc16(d)) {
        if (hbf16(d))
            BLK_RUN(mdvit_mlp_rc16_fwd_hbf16(sv.cur2, d.fc1_p, d.fc1_b, d.fc2_p, d.fc2_b, sv.x2, d.rowscale2, N_tok, sv.h, y, M, C, Hd, d.drop_p, d.key_fc1[0],
                                             d.key_fc1[1], d.key_fc2[0], d.key_fc2[1], seed, s));
        else
            BLK_RUN(mdvit_mlp_rc16_fwd(sv.cur2, d.fc1_p, d.fc1_b, d.fc2_p, d.fc2_b, sv.x2, d.rowscale2, N_tok, sv.h, y, M, C, Hd, d.drop_p, d.key_fc1[0], d.key_fc1[1],
                                       d.key_fc2[0], d.key_fc2[1], seed, s));
    } else {
        MdvitGemmDesc g;
        gemm_init(g, d);
        g.A = sv.cur2; g.B = d.fc1_w; g.M = M; g.N = Hd; g.K = C; g.lda = C; g.ldb = C; g.ldc = Hd; g.bias = d.fc1_b; g.epi = MDVIT_EPI_GELU_DUAL;
        g.e_drop_p = d.drop_p; g.e_key0 = d.key_fc1[0]; g.e_key1 = d.key_fc1[1]; g.drop_seed = seed;
        if (mode == MLP_RECOMP) { g.C = sv.h; g.C2 = nullptr; }            // gelu(u) only: the backward recomputes u
        else { g.C = sv.u; g.C2 = sv.h; }
        g.trans_a = 0; g.trans_b = 1; g.allow_split = 0;
        if (ph_takes(d, d.fc1_p, M, Hd, C, 0)) { const int rc = gemm_planes_nt(A, g, d.fc1_p, s); if (rc != MDVIT_OK) return rc; }
        else BLK_RUN(mdvit_gemm_f32(&g, s));
        gemm_init(g, d);
        g.A = sv.h; g.B = d.fc2_w; g.C = y; g.M = M; g.N = C; g.K = Hd; g.lda = Hd; g.ldb = Hd; g.ldc = C; g.bias = d.fc2_b;
        g.e_drop_p = d.drop_p; g.e_key0 = d.key_fc2[0]; g.e_key1 = d.key_fc2[1]; g.e_rowscale = d.rowscale2; g.e_rows_per_scale = N_tok;
        g.residual = sv.x2; g.ldr = C; g.drop_seed = seed;
        g.trans_a = 0; g.trans_b = 1; g.allow_split = 0;
        if (ph_takes(d, d.fc2_p, M, C, Hd, 1)) { const int rc = gemm_planes_nt(A, g, d.fc2_p, s); if (rc != MDVIT_OK) return rc; }
        else BLK_RUN(mdvit_gemm_f32(&g, s));
    }
    return MDVIT_OK;
}

// A: temporaries only the main stream touches (the caller may free them when the call returns: stream-ordered reuse); S: everything a
// side-stream kernel reads or writes (must stay alive until the side stream has finished)
int block_bwd(const MdvitBlockDesc& d, const MdvitBlockGrads& G, const MdvitBlockStreams& st, const float* x, const float* dy, float* dx, Arena& SV, Arena& A, Arena& S) {
    const int B = d.B, H = d.H, W = d.W, C = d.C, Hd = d.hidden;
    const long T = (long)B * H * W;
    const int M = (int)T, N_tok = H * W;
    const int mode = mlp_mode(d);
    const bool dgrad_only = G.dgrad_only != 0, want_w = !dgrad_only;
    const int acc = G.accumulate;
    hipStream_t s = (hipStream_t)st.main, side = s;
    Saved sv;
    layout_saved(d, SV, sv);
    if (sdpa_kind(d)) sv.x1 = const_cast<float*>(x);
    const uint32_t* seed = d.drop_p > 0.f ? d.drop_seed : nullptr;
    const bool fast_ln = C == 64 || C == 128 || C == 320 || C == 512;

    const bool have_side = st.side != nullptr && st.side != st.main;
    // Second stages of the parameter-gradient reductions (LayerNorm dgamma / dbeta, the fc2 bias column sums): ~20 us each of latency-bound work on four
    // workgroups.  When the gradients ACCUMULATE into buckets nobody reads them before the side stream is joined, so those passes run there (their
    // partial rows live in the side arena) instead of sitting on the data-gradient chain of the main stream.
    const bool defer = want_w && acc != 0 && G.ln_accumulate != 0 && have_side && fast_ln;
    struct Late { const float* part; int batches, nblk, n0; float* out0; int n1; float* out1; };
    Late late[4];
    int n_late = 0;
    if (want_w && acc == 0) {       // bias gradients ride on column sums that ACCUMULATE: clear the fresh buffers first
        const MdvitZeroItem z[4] = {{G.qkv_b, sizeof(float) * 3 * C}, {G.proj_b, sizeof(float) * C}, {mode == MLP_RC ? nullptr : G.fc1_b, sizeof(float) * Hd},
                                    {mode == MLP_RC ? nullptr : G.fc2_b, sizeof(float) * C}};
        BLK_RUN(mdvit_zero_many(z, 4, s));
    }

    // ---- MLP --------------------------------------------------------------------------------------------------------------------
    const bool masked = d.drop_p > 0.f || d.rowscale2 != nullptr;
    float* gm2 = masked ? S.take(T * C) : const_cast<float*>(dy);
    float* tmp = A.take(T * C);              // dcur2, then datt, then dcur1: consecutive lifetimes on the main stream
    float* dcur2 = tmp;
    const float* dcur2_b = nullptr;          // the second partial of dcur2 (the fused MLP backward's role 1), added by the LayerNorm backward
    float* du = nullptr;
    if (mode == MLP_RC) {
        if (masked || want_w) {
            const size_t pb = want_w ? mdvit_partials_ws_bytes(C) : 0;
            void* pw = want_w ? (defer ? S.take_bytes(pb) : A.take_bytes(pb)) : nullptr;
            if (want_w && defer) {
                int nb = 0;
                BLK_RUN(mdvit_colsum_parts(dy, C, masked ? gm2 : nullptr, pw, pb, M, C, d.drop_p, d.key_fc2[0], d.key_fc2[1], d.rowscale2, N_tok, seed, s, &nb));
                late[n_late++] = Late{(const float*)pw, 1, nb, C, G.fc2_b, 0, nullptr};
            } else {
                BLK_RUN(mdvit_colsum_f32(dy, C, want_w ? G.fc2_b : nullptr, masked ? gm2 : nullptr, pw, pb, M, C, d.drop_p, d.key_fc2[0], d.key_fc2[1], d.rowscale2, N_tok,
                                         acc, seed, s));
            }
        }
        // Full sweep: the WHOLE MLP backward from one evaluation of u, d and the activation (mdvit_mlp_rc_bwd, round 5) -- dx arrives as one partial per 256-wide hidden role
        // and the LayerNorm backward below adds them while it reads them.  (With a side stream the separate weight-gradient kernel would overlap the main stream's chain and
        // the one kernel lengthens that chain -- but it removes more work than it serialises: g_blk_mlp_bwd above.)
        const bool fuse_bwd = want_w && (Hd == 256 || Hd == 512) && (g_blk_mlp_bwd == 1 || (g_blk_mlp_bwd == 2 && !have_side));
        if (fuse_bwd) {
            const int roles = Hd / 256;
            float* parts = roles == 1 ? dcur2 : A.take((long)roles * T * C);
            const size_t wb = mdvit_mlp_rc_wgrad_ws_bytes(M, C, Hd);
            void* ww = A.take_bytes(wb);
            BLK_RUN(mdvit_mlp_rc_bwd(gm2, sv.cur2, d.fc1_p, d.fc1_b, d.fc2t_p, d.fc1t_p, parts, G.fc1_w, G.fc1_b, G.fc2_w, ww, wb, M, C, Hd, d.drop_p, d.key_fc1[0],
                                     d.key_fc1[1], seed, acc, s));
            dcur2 = parts;
            dcur2_b = roles == 2 ? parts + T * C : nullptr;
        } else {
        BLK_RUN(mdvit_mlp_rc_dgrad(gm2, sv.cur2, d.fc1_p, d.fc1_b, d.fc2t_p, d.fc1t_p, dcur2, M, C, Hd, d.drop_p, d.key_fc1[0], d.key_fc1[1], seed, s));
        if (want_w) {
            const size_t wb = mdvit_mlp_rc_wgrad_ws_bytes(M, C, Hd);
            void* ww = S.take_bytes(wb);
            if (!A.dry) { const int rc = fork_side(d, st, side); if (rc != MDVIT_OK) return rc; }
            BLK_RUN(mdvit_mlp_rc_wgrad(gm2, sv.cur2, d.fc1_p, d.fc1_b, d.fc2t_p, G.fc1_w, G.fc1_b, G.fc2_w, ww, wb, M, C, Hd, d.drop_p, d.key_fc1[0], d.key_fc1[1], seed,
                                       acc, side));
        }
        }
    } else {
        if (masked) BLK_RUN(mdvit_colsum_f32(dy, C, nullptr, gm2, nullptr, 0, M, C, d.drop_p, d.key_fc2[0], d.key_fc2[1], d.rowscale2, N_tok, 0, seed, s));
        MdvitGemmDesc g;
        int rc = MDVIT_OK;
        const bool hb = hbf16(d);
        if (hb && !(d.fc2t_p && d.fc1t_p)) return mdvit_set_error(MDVIT_E_SHAPE, "block_bwd: store_bf16 needs the transposed fc weight planes (the 16-token MLP backward)");
        if (mlp_rc16(d) && d.fc2t_p && d.fc1t_p) {
            // u recomputed, du = (gm W2) * gelu'(u) * mask and dcur2 = du W1 in one kernel; du reaches HBM only for the weight gradients
            du = want_w ? S.take(hb ? (T * Hd + 1) / 2 : T * Hd) : nullptr;
            if (hb) BLK_RUN(mdvit_mlp_rc16_dgrad_hbf16(gm2, sv.cur2, d.fc1_p, d.fc1_b, d.fc2t_p, d.fc1t_p, du, dcur2, M, C, Hd, d.drop_p, d.key_fc1[0], d.key_fc1[1], seed, s));
            else BLK_RUN(mdvit_mlp_rc16_dgrad(gm2, sv.cur2, d.fc1_p, d.fc1_b, d.fc2t_p, d.fc1t_p, du, dcur2, M, C, Hd, d.drop_p, d.key_fc1[0], d.key_fc1[1], seed, s));
        } else {
            du = S.take(T * Hd);
            gemm_init(g, d);
            g.epi = MDVIT_EPI_DGELU; g.e_drop_p = d.drop_p; g.e_key0 = d.key_fc1[0]; g.e_key1 = d.key_fc1[1]; g.drop_seed = seed;
            if (mode == MLP_RECOMP) { g.rc_a = sv.cur2; g.rc_lda = C; g.rc_b = d.fc1_w; g.rc_ldb = C; g.rc_bias = d.fc1_b; g.rc_k = C; }
            else { g.gelu_u = sv.u; g.ldu = Hd; }
            g.allow_split = 0;
            rc = gemm_dgrad(A, d, g, gm2, d.fc2_w, d.fc2_wt, du, M, Hd, C, s, d.fc2t_p);  // du = (gm W2) * gelu'(u) * mask
            if (rc != MDVIT_OK) return rc;
            gemm_init(g, d);
            g.allow_split = 1;
            rc = gemm_dgrad(A, d, g, du, d.fc1_w, d.fc1_wt, dcur2, M, C, Hd, s, d.fc1t_p);     // dx = du W1
            if (rc != MDVIT_OK) return rc;
        }
        if (want_w) {
            if (!A.dry) { rc = fork_side(d, st, side); if (rc != MDVIT_OK) return rc; }
            rc = gemm_wgrad(S, d, gm2, sv.h, G.fc2_w, G.fc2_b, M, C, Hd, acc, side, 0, hb);
            if (rc == MDVIT_OK) rc = gemm_wgrad(S, d, du, sv.cur2, G.fc1_w, G.fc1_b, M, Hd, C, acc, side, hb, 0);
            if (rc != MDVIT_OK) return rc;
        }
    }
    // ---- LN2 (+ the residual branch's gradient, which is dy itself) ---------------------------------------------------------------
    // The proj Linear's masked upstream gradient gm1 = dx2 * mask * row scale leaves the same kernel (one pass over dx2 instead of two)
    const bool masked1 = d.drop_p > 0.f || d.rowscale1 != nullptr;
    float* dx2 = masked1 ? A.take(T * C) : S.take(T * C);      // (unmasked: it IS the proj weight gradient's operand)
    float* gm1 = masked1 ? S.take(T * C) : dx2;
    {
        const bool lnw = want_w || !fast_ln;
        const size_t pb = lnw ? mdvit_partials_ws_bytes(2 * C) : 0;
        void* pw = lnw ? A.take_bytes(pb) : nullptr;
        float* dg = want_w ? G.n2_g : (lnw ? A.take((long)d.ln_groups * C) : nullptr);
        float* db = want_w ? G.n2_b : (lnw ? A.take((long)d.ln_groups * C) : nullptr);
        if (!A.dry) mdvit_layernorm_bwd_next_dy2(dcur2_b);
        if (defer) {
            void* pws = S.take_bytes(pb);
            int nb = 0;
            BLK_RUN(mdvit_layernorm_bwd_parts(dcur2, sv.x2, d.n2_g, sv.mean2, sv.rstd2, dy, dx2, masked1 ? gm1 : nullptr, pws, pb, M, C, d.ln_groups, d.drop_p, d.key_proj[0],
                                              d.key_proj[1], d.rowscale1, N_tok, seed, s, &nb));
            late[n_late++] = Late{(const float*)pws, d.ln_groups, nb, C, dg, C, db};
        } else if (masked1 && fast_ln) {
            BLK_RUN(mdvit_layernorm_bwd_masked(dcur2, sv.x2, d.n2_g, sv.mean2, sv.rstd2, dy, dx2, gm1, dg, db, pw, pb, M, C, d.ln_groups, d.drop_p, d.key_proj[0],
                                               d.key_proj[1], d.rowscale1, N_tok, seed, s));
        } else {
            BLK_RUN(mdvit_layernorm_bwd(dcur2, sv.x2, d.n2_g, sv.mean2, sv.rstd2, dy, dx2, dg, db, pw, pb, M, C, d.ln_groups, s));
            if (masked1) BLK_RUN(mdvit_colsum_f32(dx2, C, nullptr, gm1, nullptr, 0, M, C, d.drop_p, d.key_proj[0], d.key_proj[1], d.rowscale1, N_tok, 0, seed, s));
        }
    }
    // ---- proj Linear: data gradient, weight gradient (side) ---------------------------------------------------------------------------
    float* datt = tmp;
    {
        MdvitGemmDesc g;
        gemm_init(g, d);
        g.allow_split = 1;
        int rc = MDVIT_OK;
        if (d.precision == 1 && (C == 64 || C == 128) && d.projt_p && M >= 1024) {
            BLK_RUN(mdvit_linear_rc(gm1, C, d.projt_p, (long)C * C, nullptr, datt, C, M, C, C, 0.f, 0, 0, nullptr, 1, nullptr, 0, nullptr, s));
        } else {
            rc = gemm_dgrad(A, d, g, gm1, d.proj_w, d.proj_wt, datt, M, C, C, s, d.projt_p);
            if (rc != MDVIT_OK) return rc;
        }
        if (want_w) {
            if (!A.dry) { rc = fork_side(d, st, side); if (rc != MDVIT_OK) return rc; }
            rc = gemm_wgrad(S, d, gm1, sv.att, G.proj_w, G.proj_b, M, C, C, acc, side);
            if (rc != MDVIT_OK) return rc;
        }
    }
    // ---- attention core + adapter ---------------------------------------------------------------------------------------------------
    float* e = !d.label ? nullptr : G.e_out ? G.e_out : A.take((long)B * C);      // (e_out: the caller runs every adapter's backward at once, mdvit_da_bwd_many)
    float* dqkv = nullptr;
    if (sdpa_kind(d)) {
        // Attention_Sup's core backward (sdpa.hip): the probabilities are recomputed from the row log-sum-exp; e = sum_n g * att for the adapter
        dqkv = S.take(T * 3 * C);
        float* delta = A.take(2L * B * d.heads * N_tok);
        BLK_RUN(mdvit_sdpa_mfma_bwd(tmp, sv.qkv, sv.lse, sv.att, sv.a, dqkv, e, delta, B, N_tok, C, d.heads, s));
    } else {
    const size_t fab = mdvit_factoratt_ws_bytes(B, N_tok, C, d.heads);
    void* faws = S.take_bytes(fab);            // holds dU, which the deferred window-weight gradients read
    if (dgrad_only && G.aux_first && d.label) {
        // the first adapter of the network in the data-gradient-only sweep: e = sum_n g * att alone, the adapter's (negated) gradient, and
        // nothing is handed on -- nothing below carries an adapter (ops._FactorAtt, aux_first)
        BLK_RUN(mdvit_factoratt_bwd(datt, sv.qkv, sv.att, sv.U, d.w3, d.b3, d.w5, d.b5, d.w7, d.b7, sv.a, sv.kmax, sv.ksum, sv.Mmat, nullptr, e, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, faws, fab, B, H, W, C, d.heads, d.s3, d.s5, d.s7, s));
        if (!G.e_out) {
            const size_t dab = mdvit_da_ws_bytes(B, d.da_hidden, C);
            void* daws = A.take_bytes(dab);
            BLK_RUN(mdvit_da_bwd(d.label, d.da_w1, d.da_b1, d.da_w2, d.da_b2, sv.a, e, -1.0f, G.da_w1, G.da_b1, G.da_w2, G.da_b2, daws, dab, B, d.D, d.da_hidden, C, d.heads, s));
        }
        return MDVIT_OK;
    }
    dqkv = S.take(T * 3 * C);
    {
        // window-weight gradients: deferred to the side stream when they accumulate into buckets (they read dU in the SAME workspace and v);
        // otherwise produced by the backward call itself, into the fresh buffers
        const bool deferred = want_w && acc && have_side;
        const bool inl = want_w && !deferred;
        BLK_RUN(mdvit_factoratt_bwd(datt, sv.qkv, sv.att, sv.U, d.w3, d.b3, d.w5, d.b5, d.w7, d.b7, sv.a, sv.kmax, sv.ksum, sv.Mmat, dqkv, e, inl ? G.w3 : nullptr,
                                    inl ? G.b3 : nullptr, inl ? G.w5 : nullptr, inl ? G.b5 : nullptr, inl ? G.w7 : nullptr, inl ? G.b7 : nullptr, faws, fab, B, H, W, C,
                                    d.heads, d.s3, d.s5, d.s7, s));
        if (deferred) {
            if (!A.dry) { const int rc = fork_side(d, st, side); if (rc != MDVIT_OK) return rc; }
            BLK_RUN(mdvit_factoratt_wgrad(sv.qkv, faws, fab, G.w3, G.b3, G.w5, G.b5, G.w7, G.b7, B, H, W, C, d.heads, d.s3, d.s5, d.s7, 1, side));
        }
        if (inl && acc) return mdvit_set_error(MDVIT_E_SHAPE, "block_bwd: accumulating window-weight gradients need the side stream");
    }
    }
    if (d.label && !G.e_out) {
        const size_t dab = mdvit_da_ws_bytes(B, d.da_hidden, C);
        void* daws = A.take_bytes(dab);
        BLK_RUN(mdvit_da_bwd(d.label, d.da_w1, d.da_b1, d.da_w2, d.da_b2, sv.a, e, dgrad_only ? -1.0f : 1.0f, G.da_w1, G.da_b1, G.da_w2, G.da_b2, daws, dab, B, d.D,
                             d.da_hidden, C, d.heads, s));
    }
    // ---- qkv Linear -----------------------------------------------------------------------------------------------------------------
    float* dcur1 = tmp;
    {
        MdvitGemmDesc g;
        gemm_init(g, d);
        g.allow_split = 1;
        int rc = gemm_dgrad(A, d, g, dqkv, d.qkv_w, d.qkv_wt, dcur1, M, C, 3 * C, s, d.qkvt_p);
        if (rc != MDVIT_OK) return rc;
        if (want_w) {
            if (!A.dry) { rc = fork_side(d, st, side); if (rc != MDVIT_OK) return rc; }
            rc = gemm_wgrad(S, d, dqkv, sv.cur1, G.qkv_w, G.qkv_b, M, 3 * C, C, acc, side);
            if (rc != MDVIT_OK) return rc;
        }
    }
    // ---- LN1 (+ dx2 along the residual branch) ------------------------------------------------------------------------------------
    float* dx1 = (sdpa_kind(d) && dx) ? dx : S.take(T * C);          // (Block_adapt: no ConvPosEnc below -- this IS the block's input gradient)
    {
        const bool lnw = want_w || !fast_ln;
        const size_t pb = lnw ? mdvit_partials_ws_bytes(2 * C) : 0;
        void* pw = lnw ? A.take_bytes(pb) : nullptr;
        float* dg = want_w ? G.n1_g : (lnw ? A.take((long)d.ln_groups * C) : nullptr);
        float* db = want_w ? G.n1_b : (lnw ? A.take((long)d.ln_groups * C) : nullptr);
        if (defer) {
            void* pws = S.take_bytes(pb);
            int nb = 0;
            BLK_RUN(mdvit_layernorm_bwd_parts(dcur1, sv.x1, d.n1_g, sv.mean1, sv.rstd1, dx2, dx1, nullptr, pws, pb, M, C, d.ln_groups, 0.f, 0, 0, nullptr, 1, nullptr, s, &nb));
            late[n_late++] = Late{(const float*)pws, d.ln_groups, nb, C, dg, C, db};
        } else {
            BLK_RUN(mdvit_layernorm_bwd(dcur1, sv.x1, d.n1_g, sv.mean1, sv.rstd1, dx2, dx1, dg, db, pw, pb, M, C, d.ln_groups, s));
        }
    }
    if (n_late > 0 && !A.dry) {          // one fork for the block's deferred second stages: they ACCUMULATE into the LayerNorm / bias buckets
        const int rcf = fork_side(d, st, side);
        if (rcf != MDVIT_OK) return rcf;
        for (int i = 0; i < n_late; ++i) {
            const Late& L = late[i];
            const int rc = mdvit_reduce_partials_batched2_acc(L.part, L.batches, L.nblk, L.n0, L.out0, L.n1, L.out1, 1, side);
            if (rc != MDVIT_OK) return rc;
        }
    }
    // ---- ConvPosEnc -----------------------------------------------------------------------------------------------------------------
    if (sdpa_kind(d)) return MDVIT_OK;
    if (dx) BLK_RUN(mdvit_dwconv3x3_bwd(dx1, x, d.cpe_w, dx, nullptr, nullptr, nullptr, 0, B, H, W, C, 1, 1, 0, s));
    if (want_w) {
        const size_t pb = mdvit_partials_ws_bytes(10 * C);
        void* pw = S.take_bytes(pb);
        if (acc && !A.dry) { const int rc = fork_side(d, st, side); if (rc != MDVIT_OK) return rc; }
        BLK_RUN(mdvit_dwconv3x3_bwd(dx1, x, d.cpe_w, nullptr, G.cpe_w, G.cpe_b, pw, pb, B, H, W, C, 1, 1, acc, acc ? side : s));
    }
    return MDVIT_OK;
}

int check_desc(const MdvitBlockDesc& d, const char* what) {
    MDVIT_CHECK_ARG(d.B > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.heads > 0 && d.C % d.heads == 0 && d.hidden > 0 && d.C % 4 == 0, MDVIT_E_SHAPE,
                    "%s: bad geometry B=%d H=%d W=%d C=%d heads=%d hidden=%d", what, d.B, d.H, d.W, d.C, d.heads, d.hidden);
    MDVIT_CHECK_ARG(d.precision == 0 || d.precision == 1, MDVIT_E_SHAPE, "%s: precision %d (0: fp32, 1: bf16x3)", what, d.precision);
    MDVIT_CHECK_ARG(d.ln_groups >= 1 && ((long)d.B * d.H * d.W) % d.ln_groups == 0, MDVIT_E_SHAPE, "%s: %d LayerNorm groups do not divide the rows", what, d.ln_groups);
    MDVIT_CHECK_ARG(d.attn_kind == 0 || d.attn_kind == 1, MDVIT_E_SHAPE, "%s: attn_kind %d (0: SerialBlock_adapt, 1: the DeiT Block_adapt)", what, d.attn_kind);
    if (d.attn_kind == 1)
        MDVIT_CHECK_ARG(d.H * d.W == 256 && d.heads <= 6 && d.C == 64 * d.heads && d.drop_p == 0.f && !d.rowscale1 && !d.rowscale2 && !d.store_bf16, MDVIT_E_SHAPE,
                        "%s: attn_kind 1 is built for 256 tokens, head dimension 64, <= 6 heads, no dropout (H=%d W=%d C=%d heads=%d)", what, d.H, d.W, d.C, d.heads);
    MDVIT_CHECK_ARG((d.attn_kind == 1 || (d.cpe_w && d.cpe_b && d.w3 && d.b3 && d.w5 && d.b5 && d.w7 && d.b7)) && d.n1_g && d.n1_b && d.qkv_w && d.proj_w && d.proj_b && d.n2_g &&
                        d.n2_b && d.fc1_w && d.fc1_b && d.fc2_w && d.fc2_b, MDVIT_E_SHAPE, "%s: null parameter", what);
    MDVIT_CHECK_ARG(!d.label || (d.da_w1 && d.da_b1 && d.da_w2 && d.da_b2 && d.D > 0 && d.da_hidden > 0), MDVIT_E_SHAPE, "%s: a domain label needs the adapter's parameters", what);
    return MDVIT_OK;
}

}  // namespace

extern "C" size_t mdvit_block_save_bytes(const MdvitBlockDesc* d) {
    if (!d || check_desc(*d, "block_save_bytes") != MDVIT_OK) return 0;
    Arena SV{nullptr, 0, 0, true};
    Saved sv;
    layout_saved(*d, SV, sv);
    return SV.off;
}

extern "C" size_t mdvit_block_fwd_ws_bytes(const MdvitBlockDesc* d) {
    if (!d || check_desc(*d, "block_fwd_ws_bytes") != MDVIT_OK) return 0;
    Arena SV{nullptr, 0, 0, true}, A{nullptr, 0, 0, true};
    if (block_fwd(*d, nullptr, nullptr, SV, A, nullptr) != MDVIT_OK) return 0;
    return A.off + 256;
}

extern "C" int mdvit_block_fwd(const MdvitBlockDesc* d, const float* x, float* y, void* save, size_t save_bytes, void* ws, size_t ws_bytes, void* stream) {
    MDVIT_CHECK_ARG(d && x && y && save, MDVIT_E_SHAPE, "block_fwd: null argument");
    const int rc = check_desc(*d, "block_fwd");
    if (rc != MDVIT_OK) return rc;
    MDVIT_CHECK_ARG(aligned16(x) && aligned16(y) && (reinterpret_cast<uintptr_t>(save) & 255) == 0 && (!ws || (reinterpret_cast<uintptr_t>(ws) & 255) == 0), MDVIT_E_ALIGN,
                    "block_fwd: x / y must be 16-byte, save / ws 256-byte aligned");
    MDVIT_CHECK_ARG(save_bytes >= mdvit_block_save_bytes(d), MDVIT_E_WORKSPACE, "block_fwd: save buffer too small: need %zu bytes (mdvit_block_save_bytes), got %zu",
                    mdvit_block_save_bytes(d), save_bytes);
    const size_t need = mdvit_block_fwd_ws_bytes(d);
    MDVIT_CHECK_ARG(ws_bytes >= need && ws, MDVIT_E_WORKSPACE, "block_fwd: workspace too small: need %zu bytes (mdvit_block_fwd_ws_bytes), got %zu", need, ws_bytes);
    Arena SV{(char*)save, 0, save_bytes, false}, A{(char*)ws, 0, ws_bytes, false};
    return block_fwd(*d, x, y, SV, A, (hipStream_t)stream);
}

extern "C" size_t mdvit_block_bwd_ws_bytes(const MdvitBlockDesc* d, const MdvitBlockGrads* g, int32_t with_side_stream, size_t* side_bytes) {
    if (side_bytes) *side_bytes = 0;
    if (!d || !g || check_desc(*d, "block_bwd_ws_bytes") != MDVIT_OK) return 0;
    Arena SV{nullptr, 0, 0, true}, A{nullptr, 0, 0, true}, S{nullptr, 0, 0, true};
    MdvitBlockStreams st;
    memset(&st, 0, sizeof(st));
    st.main = (void*)1; st.side = with_side_stream ? (void*)2 : (void*)1;          // only compared, never used: the run is dry
    float dummy = 0.f;
    if (block_bwd(*d, *g, st, nullptr, &dummy, &dummy, SV, A, S) != MDVIT_OK) return 0;
    if (side_bytes) *side_bytes = S.off + 256;
    return A.off + 256;
}

extern "C" int mdvit_block_bwd(const MdvitBlockDesc* d, const MdvitBlockGrads* g, const MdvitBlockStreams* st, const float* x, const void* save, size_t save_bytes,
                               const float* dy, float* dx, void* ws, size_t ws_bytes, void* ws_side, size_t ws_side_bytes) {
    MDVIT_CHECK_ARG(d && g && st && x && save && dy && ws && ws_side, MDVIT_E_SHAPE, "block_bwd: null argument");
    mdvit_layernorm_bwd_next_dy2(nullptr);       // (a call that failed between announcing a second dy partial and the LayerNorm backward must not leave it behind)
    const int rc = check_desc(*d, "block_bwd");
    if (rc != MDVIT_OK) return rc;
    MDVIT_CHECK_ARG(aligned16(x) && aligned16(dy) && (!dx || aligned16(dx)) && (reinterpret_cast<uintptr_t>(save) & 255) == 0 && (reinterpret_cast<uintptr_t>(ws) & 255) == 0 &&
                        (reinterpret_cast<uintptr_t>(ws_side) & 255) == 0, MDVIT_E_ALIGN, "block_bwd: x / dy / dx must be 16-byte, save / ws 256-byte aligned");
    MDVIT_CHECK_ARG(save_bytes >= mdvit_block_save_bytes(d), MDVIT_E_WORKSPACE, "block_bwd: save buffer too small");
    size_t need_side = 0;
    const size_t need = mdvit_block_bwd_ws_bytes(d, g, st->side != nullptr && st->side != st->main, &need_side);
    MDVIT_CHECK_ARG(ws_bytes >= need && ws_side_bytes >= need_side, MDVIT_E_WORKSPACE,
                    "block_bwd: workspace too small: need %zu + %zu bytes (mdvit_block_bwd_ws_bytes), got %zu + %zu", need, need_side, ws_bytes, ws_side_bytes);
    if (!g->dgrad_only) {
        MDVIT_CHECK_ARG((d->attn_kind == 1 || (g->cpe_w && g->cpe_b && g->w3 && g->b3 && g->w5 && g->b5 && g->w7 && g->b7)) && g->n1_g && g->n1_b && g->qkv_w && g->proj_w &&
                            g->proj_b && g->n2_g && g->n2_b && g->fc1_w && g->fc1_b && g->fc2_w && g->fc2_b && (!d->qkv_b || g->qkv_b), MDVIT_E_SHAPE,
                        "block_bwd: null gradient output");
    }
    MDVIT_CHECK_ARG(!d->label || g->e_out || (g->da_w1 && g->da_b1 && g->da_w2 && g->da_b2), MDVIT_E_SHAPE, "block_bwd: the adapter's gradient outputs are missing");
    MDVIT_CHECK_ARG(d->precision == 0 || (d->qkv_wt && d->proj_wt && (mlp_mode(*d) == MLP_RC || (mlp_rc16(*d) && d->fc2t_p && d->fc1t_p) || (d->fc1_wt && d->fc2_wt))), MDVIT_E_SHAPE,
                    "block_bwd: the bf16x3 data-gradient GEMMs need the transposed weights");
    Arena SV{(char*)const_cast<void*>(save), 0, save_bytes, false}, A{(char*)ws, 0, ws_bytes, false}, S{(char*)ws_side, 0, ws_side_bytes, false};
    return block_bwd(*d, *g, *st, x, dy, dx, SV, A, S);
}
