/* libmdvit_hip.so -- C ABI of the MI355X-native MDViT forward/backward path.
 *
 * Conventions:
 *  - every function ENQUEUES work on `stream` (a hipStream_t passed as void*) and returns
 *    immediately; no internal device synchronisation; no allocation -- the caller owns every
 *    buffer including workspaces; pointers are borrowed for the duration of the enqueue.
 *  - return 0 (MDVIT_OK) or an MDVIT_E_* code; the message is in mdvit_last_error() (thread-local).
 *  - all tensors are fp32, activations in token-major NHWC: [B, H*W, C] row-major ("tokens").
 *    Weights keep the layouts PyTorch's modules store (Linear [out,in], Conv [out,in/g,kh,kw]),
 *    so a reference state_dict is consumed without repacking.
 *  - file:line citations are into the reference tree (siyi-wind/MDViT) the entry point replaces.
 */
#ifndef MDVIT_HIP_H
#define MDVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDVIT_ABI_VERSION 1

enum {
    MDVIT_OK = 0,
    MDVIT_E_SHAPE = 1,
    MDVIT_E_DTYPE = 2,
    MDVIT_E_ALIGN = 3,
    MDVIT_E_WORKSPACE = 4,
    MDVIT_E_HIP = 5
};

enum { MDVIT_EPI_NONE = 0, MDVIT_EPI_GELU_DUAL = 1, MDVIT_EPI_DGELU = 2 };
enum { MDVIT_ACT_NONE = 0, MDVIT_ACT_HSWISH = 1, MDVIT_ACT_RELU = 2 };

const char* mdvit_last_error(void);
int mdvit_version(void);

/* Measurement only (bench.py's `roofline`; no reference counterpart): HIP events owned by the library, and "arm": the NEXT GEMM
 * main-kernel launch (mdvit_gemm_f32 incl. the implicit convolution and the weight-gradient kernel) records its own begin / end
 * timestamps into (start, stop) -- the dispatch's execution time as rocprofv3 --kernel-trace reports it.  arm(NULL, NULL) disarms. */
int mdvit_event_create(void** out_event);
int mdvit_event_destroy(void* event);
int mdvit_event_elapsed_ms(void* start_event, void* stop_event, float* ms);
int mdvit_timing_arm(void* start_event, void* stop_event);

/* Reductions over the token axis (weight / bias gradients, column sums) write ONE row of partial sums per workgroup into
 * the caller's workspace and add the rows in a fixed order in a second tiny launch -- deterministic, and much faster than
 * ~1000 same-address float atomics.  `ws` of those entry points: mdvit_partials_ws_bytes(n) bytes, n = number of floats
 * the reduction produces (the entry point's comment says which). */
size_t mdvit_partials_ws_bytes(int32_t n_outputs);

/* ---- GEMM family -------------------------------------------------------------------------
 * C[M,N] = op(A)[M,K] * op(B)[K,N] (+bias[N]) with fused neighbours.
 *   trans_a = 0: A is [M,K] row-major (lda);   1: A is stored [K,M] row-major (wgrad, A = dY^T).
 *   trans_b = 1: B is [N,K] row-major (an nn.Linear / 1x1-conv weight); 0: B is [K,N] row-major.
 * (The backward of `res + DropPath(Dropout(.))`, mdvit.py:311,354 / mpvit.py:73-78, masks the upstream gradient once
 *  with mdvit_colsum_f32 and feeds the masked tensor to the dgrad / wgrad GEMMs.)
 * Epilogues:
 *   NONE      : C = acc + bias; then optional dropout(e_drop_p), row scale (DropPath), + residual.
 *   GELU_DUAL : C = acc + bias (pre-activation u), C2 = dropout(gelu_erf(u))       (mpvit.py:73-75)
 *   DGELU     : C = acc * gelu'(gelu_u) * dropmask(e_key)                          (backward of the above)
 * allow_split: the reduction may be split across workgroups: every split writes a dense [M,N] slab into `ws`,
 * a second kernel adds the slabs in a fixed order (deterministic; no atomics).
 * Replaces: nn.Linear mdvit.py:288,310  mpvit.py:73,76; 1x1 nn.Conv2d mdvit.py:98,589
 *           Decoders.py:185,300-311; their autograd backward. */
typedef struct MdvitGemmDesc {
    const float* A; const float* B; float* C; float* C2;
    int64_t lda, ldb, ldc;
    int32_t M, N, K;
    int32_t trans_a, trans_b;
    const float* bias;
    int32_t epi;
    float e_drop_p; uint32_t e_key0, e_key1;
    const float* e_rowscale; int32_t e_rows_per_scale;
    const float* residual; int64_t ldr;
    const float* gelu_u; int64_t ldu;
    int32_t allow_split;
    void* ws; uint64_t ws_bytes;      /* scratch for split reductions: mdvit_gemm_ws_bytes(desc) (0 = none needed) */
    int32_t accumulate;               /* C += result (gradient accumulation straight into a persistent buffer) */
    float* colsum_a;                  /* TN only, optional: colsum_a[m] += sum_k A[k][m] -- the bias gradient (column sums of dY) taken
                                       * from the wgrad's own A stream instead of a second pass over dY; the caller zeroes it
                                       * (or passes a gradient bucket to accumulate into) */
    int32_t precision;                /* 0: fp32 MFMA (bit-for-bit an fmaf chain).  1: "bf16x3" -- operands split hi+lo into two bf16
                                       * planes while staged, hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32
                                       * accumulation (~1e-5 relative); built for NT (all epilogues) and TN (plain) */
    const uint32_t* drop_seed;        /* optional device {s0,s1}: effective keys (key0 ^ s0, key1 + s1) -- lets a captured
                                         HIP graph draw fresh dropout masks on every replay */
    /* DGELU with a RECOMPUTED pre-activation (NT only, gelu_u == NULL): u = rc_a[M,rc_k] rc_b[N,rc_k]^T + rc_bias[N] is formed
     * in the kernel as a second product over the output tile, with the slab / MFMA sequence of the forward GEMM (bit-identical
     * u).  For the MLP of the C <= 128 stages (Mlp.fc1, mpvit.py:73-76: K = C): the forward then stores gelu(u) only
     * (GELU_DUAL with C2 == NULL writes gelu(u) x dropout to C) and neither pass moves the [tokens, hidden] u through HBM. */
    const float* rc_a; int64_t rc_lda; const float* rc_b; int64_t rc_ldb; const float* rc_bias; int32_t rc_k;
    /* Implicit 3x3 convolution (conv_c > 0; NT, precision >= 1, plain epilogue): A is NOT a matrix but the NHWC image
     * [M / (conv_ho conv_wo), conv_h, conv_w, conv_c]; row m = output pixel (b, ho, wo), K = 9 conv_c ordered (tap, channel):
     * A[m][tap * conv_c + c] = image[b, ho * stride + (tap / 3 - 1) * dilation, wo * stride + (tap % 3 - 1) * dilation, c] (0 outside),
     * gathered by the kernel while it stages the operand -- the [M, 9 C] im2col matrix is never written (nn.Conv2d(k = 3, padding =
     * dilation) of mdvit.py:557-564, Utils/_deeplab.py:115-122, torchvision's ResNet BasicBlock; its data gradient at stride 1 is the
     * same call on the output gradient with the flipped, transposed weight).  B = the weight as [N][tap][channel]
     * (mdvit_conv_weight_relayout).  conv_c % 32 == 0; lda is ignored. */
    int32_t conv_c, conv_h, conv_w, conv_ho, conv_wo, conv_stride, conv_dilation;
    /* conv_up > 1 (NT only): the image is read as if zero-upsampled by conv_up -- tap position p maps to source pixel p / conv_up when
     * divisible, else contributes 0: the data gradient of a stride-conv_up convolution (a transposed convolution) as the same
     * implicit GEMM over the INPUT pixels (conv_ho x conv_wo), image = dy (conv_h x conv_w), conv_stride = 1.  0 / 1 = off.
     * conv_up == 2 with conv_ho == 2 conv_h, conv_wo == 2 conv_w and M / 4 a multiple of 256: the kernel deals its tiles by the parity class (y % 2, x % 2)
     * of the input pixels and walks only the class's 1 / 2 / 2 / 4 live taps (an internal schedule: same rows of C, same sums without their zero terms;
     * MDVIT_CONV_PHASE=0 keeps the nine-tap walk). */
    int32_t conv_up;
    /* TN, precision 1 only: that operand is stored as bf16 ([K, M] / [K, N] of 2-byte elements behind the float pointer, leading dimension in elements,
     * % 4 == 0): the saved hidden activations of the "mixed" mode (MdvitBlockDesc.store_bf16).  It enters the product as its single bf16 plane (two
     * MFMAs per product instead of three); at most one of the two. */
    int32_t a_bf16, b_bf16;
    /* TN with conv_c > 0 (the weight gradient of a 3x3 convolution): write / accumulate C in the PyTorch layout [M = Cout][conv_c][3][3] (ldc = 9 conv_c) instead of the
     * tap-major [M][9][conv_c] the product is formed in -- the relayout pass of round 2-3 (one launch and one round trip of the gradient per convolution) folded into the
     * kernel's / the slab reduction's stores. */
    int32_t conv_wgrad_nchw;
} MdvitGemmDesc;
size_t mdvit_gemm_ws_bytes(const MdvitGemmDesc* desc);
/* The same plain product (desc: shape, layout, precision of ONE group; its A / B / C, bias, epilogue and allow_split are ignored / must be off) on G <= 8 operand
 * triples in ONE launch.  Replaces the G x 4 separate `W_fuse[:, block] @ W_linear` products (and their two gradient products each) with which the peer heads'
 * 1x1 convolutions are composed (Decoders.py:315-339: linear_c -> interpolate -> cat -> linear_fuse, evaluated as resize((W_fuse,q W_q) x_q) by decode.py). */
#define MDVIT_GEMM_MAX_GROUPS 8
int mdvit_gemm_f32_grouped(const MdvitGemmDesc* desc, int32_t G, const void* const* A, const void* const* B, void* const* C, void* stream);
/* ... NN / NT with a bias vector per group: the peer heads' low-resolution linear_c products (Decoders.py:319-328) of all G heads in one launch. */
int mdvit_gemm_f32_grouped_bias(const MdvitGemmDesc* desc, int32_t G, const void* const* A, const void* const* B, void* const* C, const void* const* bias, void* stream);
/* The bias part of the same composition for n <= 16 (head, scale) items in one launch: out_i[r] = sum_c W_i[r][c] b_i[c] (W_i: [rows, cols] blocks of the fuse
 * weight, leading dimension ldw; b_i: the linear_c bias), and its gradients: dW_i[r][c] += dout_i[r] b_i[c] (into the block gradient the grouped product wrote,
 * leading dimension lddw; dW may be NULL), db_i[c] (+)= sum_r W_i[r][c] dout_i[r] (db may be NULL). */
int mdvit_compose_bias(int32_t n, const void* const* W, int64_t ldw, const void* const* b, void* const* out, int32_t rows, int32_t cols, void* stream);
int mdvit_compose_bias_bwd(int32_t n, const void* const* W, int64_t ldw, const void* const* b, const void* const* dout, void* const* dW, int64_t lddw,
                           void* const* db, int32_t db_accumulate, int32_t rows, int32_t cols, void* stream);
/* which kernel variant the launch will use (tile BMxBN, number of K splits): for profiling / roofline accounting */
int mdvit_gemm_plan(const MdvitGemmDesc* desc, int32_t* tile_m, int32_t* tile_n, int32_t* splits);
/* out[cols, rows] = in[rows, cols]^T (row-major, in may be a column slice with leading dimension ld_in).  Used on WEIGHTS
 * only: with precision = 1 the data-gradient GEMM  dX = dY W  reads W^T so that both operands are k-contiguous. */
int mdvit_transpose_f32(const float* in, int64_t ld_in, float* out, int32_t rows, int32_t cols, void* stream);
/* n <= 24 transposes out_i[cols_i][rows_i] = in_i[rows_i][cols_i]^T (row-contiguous inputs, leading dimension ld_in_i) in ONE launch, the items in the kernel's
 * arguments: the W^T of the peer heads' composed weights (per-step temporaries that every data-gradient product of both sweeps reads). */
int mdvit_transpose_batch(int32_t n, const void* const* in, const int64_t* ld_in, void* const* out, const int32_t* rows, const int32_t* cols, void* stream);
/* n transposes in one launch (the per-step refresh of every cached W^T after the optimizer update).  items_dev: device array
 * [n][5] of int64 {in pointer, out pointer, ld_in, rows, cols}; blocks_per_item workgroups walk each item's 32x32 tiles. */
int mdvit_transpose_many(const void* items_dev, int32_t n, int32_t blocks_per_item, void* stream);
/* Tuning hook for tools/gemm_sweep.py: pin the tile configuration (0: 128x128, 1: 256x64, 2: 64x64; -1: planner) and the
 * requested K-split (one of the planner's candidates; 0: planner) of every following mdvit_gemm_f32 call. */
int mdvit_gemm_force_plan(int32_t cfg, int32_t splits);
/* The kernel symbol mdvit_gemm_f32 launches for this descriptor, as rocprofv3 prints it ("+splitk_reduce" appended when the K-split
 * reduction follows): measurement hook, bench.py matches its HIP-event timings against the committed profile by this name. */
int mdvit_gemm_kernel_name(const MdvitGemmDesc* d, char* out, int32_t cap);
/* The TN (weight-gradient: autograd's grad_out.t() @ input of nn.Linear / 1x1 conv) launches of mdvit_gemm_f32 run on a dedicated
 * kernel (gemm_tn.hip: k-major LDS image read through ds_read_b64_tr_b16).  enable = 0 routes them through the general template
 * again; cfg 0..3 = tile 128x128 / 128x64 / 64x128 / 64x64 (-1: planner), splits > 0 forces the K-split (A/B and sweep hook). */
int mdvit_gemm_tn_config(int32_t enable, int32_t cfg, int32_t splits);
/* Workgroup order of the same kernel over the 8 XCDs: 1 (default; also env MDVIT_TN_GRID_XCD) = the tiles of one K-split are neighbours on one XCD when a
 * split has <= 8 tiles (they read the same token rows), 0 = XCD-contiguous inside a split only, 2 = whole-grid order always, -1 = back to the environment's
 * choice.  Results do not depend on it (tile arithmetic and the slab reduction order are the same): A/B and test hook. */
int mdvit_gemm_tn_grid_order(int32_t mode);
int mdvit_gemm_f32(const MdvitGemmDesc* desc, void* stream);
/* Weight layouts of the implicit 3x3 convolution: w [Cout, Cin, 3, 3] (PyTorch) ->
 *   mode 0: out [Cout][tap][Cin]                    (forward:        y = conv(x, w))
 *   mode 1: out [Cin][8 - tap][Cout]                (data gradient: dx = conv(dy, flipped / transposed w), stride 1)
 *   mode 2 / 3: w = a gradient in the [Cout][tap][Cin] order of the implicit weight-gradient GEMM (TN with conv_c > 0: A = dy [tokens, Cout],
 *               B = the image, N = 9 Cin) -> out [Cout, Cin, 3, 3], overwritten (2) or accumulated into (3) */
int mdvit_conv_weight_relayout(const float* w, float* out, int32_t Cout, int32_t Cin, int32_t mode, void* stream);
/* modes 0 / 1 for many weights in one launch: items [n][5] int64 in device memory = {w, out, Cout, Cin, mode} (the per-step refresh of every cached layout) */
int mdvit_conv_weight_relayout_many(const void* items_dev, int32_t n, int32_t blocks_per_item, void* stream);

/* ---- "plane" GEMM family: operands pre-split into bf16 planes ------------------------------------------------
 * A plane tensor is [planes][rows][ld] bf16 (uint16 storage): plane 0 = hi = RNE bf16(x), plane 1 = lo = RNE bf16(x - hi),
 * `*_plane` = distance between the planes in elements.  planes = 2 runs the bf16x3 arithmetic of mdvit_gemm_f32 (bit-identical
 * results: same products, same order), planes = 1 is the bf16 speed mode (hi only, one MFMA per product).  No conversion
 * work is left in the main loop: slabs go HBM/L2 -> LDS by global_load_lds.  Weights are split once per optimizer step
 * (mdvit_split_planes_many: W for the forward, W^T for the data gradients), activations by their producers.
 *   NT : C[M,N] = A[M,K] B[N,K]^T  (a_f32 = 1: A is fp32 [M,K], lda in floats, split while staged)
 * The result of every epilogue can be written as fp32 (C) and / or as planes (Cp) -- the operand format of the next GEMM.
 *   NONE      : acc + bias, then optional dropout, DropPath row scale, + residual; accumulate / allow_split as mdvit_gemm_f32
 *   GELU_DUAL : U = acc + bias (optional fp32 pre-activation), result = dropout(gelu(U))
 *   DGELU     : result = acc * gelu'(u) * dropmask, u = gelu_u (fp32) or recomputed from rc_a / rc_b planes (+ rc_bias)
 * K % 32 == 0, N % 4 == 0 (other shapes: mdvit_gemm_f32).
 * Replaces the call sites of mdvit_gemm_f32: mdvit.py:288,310-311; mpvit.py:71-78; Decoders.py:185,196,300-331. */
typedef struct MdvitPlaneGemmDesc {
    const void* A; int64_t lda; int64_t a_plane; int32_t a_f32;
    const void* B; int64_t ldb; int64_t b_plane;
    int32_t planes;                   /* 2: bf16x3, 1: bf16 */
    int32_t trans;                    /* 0: NT.  (1: TN, the weight gradient: mdvit_gemm_planes_tn) */
    int32_t M, N, K;
    float* C; int64_t ldc;            /* fp32 result, optional */
    void* Cp; int64_t ldcp; int64_t c_plane;   /* plane result, optional */
    float* U; int64_t ldu_out;        /* GELU_DUAL: fp32 pre-activation, optional */
    const float* bias;
    int32_t epi;                      /* MDVIT_EPI_* */
    float e_drop_p; uint32_t e_key0, e_key1;
    const float* e_rowscale; int32_t e_rows_per_scale;
    const float* residual; int64_t ldr;
    const float* gelu_u; int64_t ldu;
    const void* rc_a; int64_t rc_lda; int64_t rc_a_plane; const void* rc_b; int64_t rc_ldb; int64_t rc_b_plane; const float* rc_bias; int32_t rc_k;
    int32_t allow_split; void* ws; uint64_t ws_bytes;
    int32_t accumulate;
    const uint32_t* drop_seed;
} MdvitPlaneGemmDesc;
size_t mdvit_gemm_planes_ws_bytes(const MdvitPlaneGemmDesc* desc);
int mdvit_gemm_planes_plan(const MdvitPlaneGemmDesc* desc, int32_t* tile_m, int32_t* tile_n, int32_t* splits);
int mdvit_gemm_planes_force_plan(int32_t cfg, int32_t splits);      /* tuning hook: cfg 0: 128x128, 1: 128x64, 2: 64x64; -1 / 0: planner */
int mdvit_gemm_planes(const MdvitPlaneGemmDesc* desc, void* stream);
/* The 256 x 256 phase-split kernels (csrc/gemm_ph.hip: eight waves, one workgroup per CU, global_load_lds ring with counted waits; same arithmetic and
 * results as the other tiles) serve the MFMA-bound NT products -- qkv / fc1 / the fc2 data gradient of the C >= 320 blocks (mdvit.py:267,307, mpvit.py:71-78),
 * the wide projections of Decoders.py:319-331.  mdvit_gemm_ph_prefers: 1 when mdvit_gemm_planes routes [M, K] x [N, K]^T to them (the share of real output in
 * the chip's rounds of 256 x 256 tiles decides; K % 32 == 0 (planes = 2) / % 64 (planes = 1), at least two K tiles).  mdvit_gemm_ph_config: -1 never, 0 that
 * rule (default), 1 whenever legal (A/B and test hook). */
int mdvit_gemm_ph_prefers(int32_t M, int32_t N, int32_t K, int32_t planes);
/* the same for a launch whose epilogue READS an [M, N] operand (epi_reads != 0: gelu_u of the fc2 data gradient -- mpvit.py:75's backward --, the residual of
 * mdvit.py:353,357-360, an accumulating C): one workgroup per CU hides no load latency, such launches are taken only when the tiles fill whole rounds */
int mdvit_gemm_ph_prefers_epi(int32_t M, int32_t N, int32_t K, int32_t planes, int32_t epi_reads);
int mdvit_gemm_ph_config(int32_t mode);
/* The same phase-split structure on a 128-row tile (csrc/gemm_pm.hip, round 5) for the MID-SIZE products of the C = 320 / 512 blocks (mdvit.py:267,307, mpvit.py:71-78
 * at 16-128 images: proj, fc2, the qkv / fc1 data gradients): 128 x 160 output tiles for N % 160 == 0 (16384 x 320 is exactly one workgroup per CU), 128 x 128 for
 * N % 128 == 0; fp32 A split while staged, two weight planes, one K range; results bit-identical to the other tiles.  mdvit_gemm_pm_prefers: the tile width
 * mdvit_gemm_planes would route [M, K] x [N, K]^T to (6: 128 x 160, 7: 128 x 128) or 0; mdvit_gemm_pm_config: -1 never, 0 by the rule (default), 1 whenever legal. */
int mdvit_gemm_pm_prefers(int32_t M, int32_t N, int32_t K, int32_t planes, int32_t a_f32);
int mdvit_gemm_pm_config(int32_t mode);
/* The K ranges (2-4, or 1: none) mdvit_gemm_planes gives that tile for a PLAIN product (no epilogue operand, allow_split, workspace from mdvit_gemm_planes_ws_bytes) whose
 * tiles alone would leave most of the chip idle -- the stage-3 data gradients at 16 images (4096 x 512 x 1536 / 2048: 128 tiles); the slabs are added in split order by
 * the split-K reduction kernel, so the result is that of the 128 x 128 plane tile with the same K ranges, bit for bit. */
int mdvit_gemm_pm_splits(int32_t M, int32_t N, int32_t K, int32_t planes, int32_t a_f32);
/* fp32 [rows, cols] (ld_in) -> planes [planes][rows][ld_out]; cols % 8 == 0 */
int mdvit_split_planes(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t plane_stride, int64_t rows, int32_t cols, int32_t planes, void* stream);
/* one tensor, any shape, optionally transposed (out = planes of in^T, [cols][rows]): non-leaf / sliced weights */
int mdvit_split_planes_t(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t plane_stride, int32_t rows, int32_t cols, int32_t transpose,
                         int32_t planes, void* stream);
/* n weight splits in one launch.  items_dev: device array [n][8] of int64 {src fp32, dst planes, ld_src, rows, cols, transpose (0/1),
 * ld_dst, plane_stride}; transpose = 1 writes the planes of the transposed matrix ([cols][rows]: the data gradients' W^T). */
int mdvit_split_planes_many(const void* items_dev, int32_t n, int32_t blocks_per_item, int32_t planes, void* stream);

/* Fused MLP forward of the C = 64 stages (Mlp.forward mpvit.py:71-78 + the block's DropPath / residual, mdvit.py:357-360):
 *   h = drop1(gelu(x W1^T + b1))  [M, hidden]  (written: the backward's operand),   y = res + rowscale * drop2(h W2^T + b2)  [M, C]
 * in one kernel that never re-reads h; same bf16x3 arithmetic and dropout keys as the two mdvit_gemm_f32 calls it replaces (bit-identical
 * h and y), so the backward -- fc2 data gradient with the recomputed pre-activation, wgrads -- is unchanged.  hidden % 64 == 0. */
int mdvit_mlp_fwd_f32(const float* x, const float* W1, const float* b1, const float* W2, const float* b2, const float* res,
                      const float* rowscale /* optional */, int32_t rows_per_scale, float* h, float* y, int32_t M, int32_t C, int32_t hidden,
                      float drop_p, uint32_t key1_0, uint32_t key1_1, uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed, void* stream);

/* tuning / diagnosis hook: token count from which the fused MLP forward uses its 128-token tile (<= 0: never); ablate != 0 switches
 * parts of the forward kernel off for timing experiments (tools/mlp_check.py) -- results are then wrong by design; 0 in production */
int mdvit_mlp_config(int32_t wide_min_tokens, int32_t ablate);

/* ... and its backward data path:  dx = ((gm W2) * gelu'(x W1^T + b1) * dropmask1) W1  in one kernel (gm = the masked upstream
 * gradient from mdvit_colsum_f32; W2t / W1t = the cached transposes, [hidden, C] and [C, hidden]).  du (optional, [M, hidden]):
 * the hidden-layer gradient for the weight-gradient GEMMs; with du == NULL (the data-gradient-only sweep) no [M, hidden] tensor
 * touches HBM at all.  Same products, slab order and dropout key as mdvit_gemm_f32(DGELU + rc_*) followed by the fc1 data-gradient GEMM. */
int mdvit_mlp_bwd_dgrad_f32(const float* gm, const float* x, const float* W1, const float* b1, const float* W2t, const float* W1t,
                            float* du /* optional */, float* dx, int32_t M, int32_t C, int32_t hidden, float drop_p,
                            uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, void* stream);

/* ---- MLP whose hidden activation never touches HBM (csrc/mlp_rc.hip; C = 64; replaces Mlp.forward mpvit.py:71-78 inside
 * SerialBlock_adapt mdvit.py:357-360 and autograd's backward of it) ------------------------------------------------------------------
 * Weights arrive as the per-step bf16 planes ([2][rows][cols], mdvit_split_planes_many): W1p = planes of W1 [hidden, C], W2p = planes of
 * W2 [C, hidden], W2tp = planes of W2^T [hidden, C], W1tp = planes of W1^T [C, hidden].  Same bf16x3 arithmetic, k order and dropout keys
 * as mdvit_mlp_fwd_f32 / mdvit_mlp_bwd_dgrad_f32 (y and dx bit-identical to them); nothing of size [M, hidden] is read or written:
 *   fwd   : y = res + rowscale * drop2(drop1(gelu(x W1^T + b1)) W2^T + b2)
 *   dgrad : dx = ((gm W2) * gelu'(x W1^T + b1) * dropmask1) W1                    (gm: the masked upstream gradient, mdvit_colsum_f32)
 *   wgrad : dW1 (+)= du^T x, db1 (+)= colsum(du), dW2 (+)= gm^T h with u, h, du RECOMPUTED per 32-token tile from x and gm (the operands
 *           of autograd's two weight-gradient products are never materialised); fixed-order partial sums in ws
 *           (mdvit_mlp_rc_wgrad_ws_bytes), accumulate != 0 adds into dW1 / db1 / dW2 (gradient buckets).  hidden % 256 == 0.
 *           (db2 = colsum(gm) comes from mdvit_colsum_f32's `out`.) */
int mdvit_mlp_rc_fwd(const float* x, const void* W1p, const float* b1, const void* W2p, const float* b2, const float* res,
                     const float* rowscale /* optional */, int32_t rows_per_scale, float* y, int32_t M, int32_t C, int32_t hidden,
                     float drop_p, uint32_t key1_0, uint32_t key1_1, uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed, void* stream);
int mdvit_mlp_rc_dgrad(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* dx,
                       int32_t M, int32_t C, int32_t hidden, float drop_p, uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, void* stream);
/* Streaming Linear for the short-K, output-heavy layers (K = 64 / 128): qkv and proj of the C = 64 / 128 stages (mdvit.py:288,310) and their data
 * gradients, the 64 / 128 -> 512 projections of the peer heads (Decoders.py:320-331).  y[M, N] = x[M, K] Wp^T (+ bias); with `residual` != NULL
 * also x dropout(key0, key1) x rowscale[row / rows_per_scale] + residual (mdvit_gemm_f32's FULL epilogue, same mask indices).  Wp: the bf16 hi / lo
 * planes [2][N][K] of the weight (or of its transpose for a data gradient; mdvit_split_planes_t), plane stride `wplane` elements.  A wave owns 32
 * tokens for the whole output row: x goes straight into MFMA operand registers, the weight planes stream through LDS, the kernel is a stream
 * of 16-byte stores.  N % 32 == 0. */
int mdvit_linear_rc(const float* x, int64_t lda, const void* Wp, int64_t wplane, const float* bias, float* y, int64_t ldc, int32_t M, int32_t N, int32_t K,
                    float drop_p, uint32_t key0, uint32_t key1, const float* rowscale, int32_t rows_per_scale, const float* residual, int64_t ldr,
                    const uint32_t* drop_seed, void* stream);
/* mdvit_linear_rc with the LayerNorm in front of it fused into its prologue (LN1 -> qkv of SerialBlock_adapt, mdvit.py:286-288,352): x is the
 * LayerNorm's INPUT [M, K] (contiguous, K = 64 / 128), gamma / beta [groups, K] (group = row / (M / groups)).  Writes mean / rstd [M] and the normalised rows
 * ln_out [M, K] (the operand of the qkv weight-gradient GEMM) and y = ln_out Wp^T + bias -- mdvit_layernorm_fwd's arithmetic sum for sum (equal results),
 * one launch and one pass over x less. */
int mdvit_linear_rc_ln(const float* x, const float* gamma, const float* beta, int32_t groups, float eps, float* mean, float* rstd, float* ln_out,
                       const void* Wp, int64_t wplane, const float* bias, float* y, int64_t ldc, int32_t M, int32_t N, int32_t K, void* stream);
/* The MLP forward on 16-token waves (16x16x32 MFMA tiles), built for C = 64 and C = 128: fc1 + GELU + Dropout + fc2 + Dropout + DropPath + residual
 * (mpvit.py:71-78 inside mdvit.py:357-360) in ONE kernel with the hidden chunk chained in registers; h != NULL also writes
 * h = drop1(gelu(x W1^T + b1)) [M, hidden] once for the fc2 weight-gradient GEMM (C = 128, where recomputing it there costs more than it saves). */
int mdvit_mlp_rc16_fwd(const float* x, const void* W1p, const float* b1, const void* W2p, const float* b2, const float* res, const float* rowscale,
                       int32_t rows_per_scale, float* h, float* y, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1,
                       uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed, void* stream);
/* The MLP forward with the LayerNorm in front of it fused into its prologue (LN2 -> Mlp of SerialBlock_adapt, mdvit.py:356-360): x2 [M, C] is the LayerNorm's
 * INPUT and the residual.  Writes mean / rstd [M], the normalised rows ln_out [M, C] (operand of the backward kernels) and y = x2 + rowscale * drop2(fc2(drop1(gelu(
 * fc1(ln_out))))).  C = 64: no [tokens, hidden] tensor (h must be NULL); C = 128: h != NULL written as by mdvit_mlp_rc16_fwd.  mdvit_layernorm_fwd's arithmetic
 * sum for sum: equal results, one launch and one pass over x2 less. */
int mdvit_mlp_rc_fwd_ln(const float* x2, const float* gamma, const float* beta, int32_t groups, float eps, float* mean, float* rstd, float* ln_out,
                        const void* W1p, const float* b1, const void* W2p, const float* b2, const float* rowscale, int32_t rows_per_scale, float* h, float* y,
                        int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, uint32_t key2_0, uint32_t key2_1,
                        const uint32_t* drop_seed, void* stream);
/* The same data gradient on 16-token waves (16x16x32 MFMA tiles), built for C = 64 and C = 128: the C = 128 stages' MLP backward data path
 * (mpvit.py:71-78 with hidden = 8 C, mdvit.py:357-360) in ONE kernel instead of the recomputing fc2 data-gradient GEMM + the fc1 data-gradient
 * GEMM.  du != NULL additionally writes the hidden-layer gradient [M, hidden] (operand of the two weight-gradient GEMMs of the full sweep);
 * du == NULL (data-gradient-only sweep) moves no [tokens, hidden] tensor.  hidden % 32 == 0. */
int mdvit_mlp_rc16_dgrad(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* du, float* dx,
                         int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, void* stream);
/* The three kernels above with the [tokens, hidden] tensor they leave for the weight-gradient GEMMs -- h = drop1(gelu(u)) of the forward, du of the backward --
 * stored as bf16 (2-byte elements behind the float pointer; what is stored is the hi plane the kernel forms for its own second product, so y / dx do not
 * change by a bit).  The "mixed" mode of BASELINE configs[3] (MdvitBlockDesc.store_bf16); consumed by mdvit_gemm_f32 with MdvitGemmDesc.a_bf16 / b_bf16. */
int mdvit_mlp_rc16_fwd_hbf16(const float* x, const void* W1p, const float* b1, const void* W2p, const float* b2, const float* res, const float* rowscale,
                             int32_t rows_per_scale, float* h, float* y, int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1,
                             uint32_t key2_0, uint32_t key2_1, const uint32_t* drop_seed, void* stream);
int mdvit_mlp_rc_fwd_ln_hbf16(const float* x2, const float* gamma, const float* beta, int32_t groups, float eps, float* mean, float* rstd, float* ln_out,
                              const void* W1p, const float* b1, const void* W2p, const float* b2, const float* rowscale, int32_t rows_per_scale, float* h, float* y,
                              int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, uint32_t key2_0, uint32_t key2_1,
                              const uint32_t* drop_seed, void* stream);
int mdvit_mlp_rc16_dgrad_hbf16(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* du, float* dx,
                               int32_t M, int32_t C, int32_t Hd, float drop_p, uint32_t key1_0, uint32_t key1_1, const uint32_t* drop_seed, void* stream);
/* The whole backward of the C = 64 MLP (mpvit.py:71-78 in mdvit.py:357-360) from ONE evaluation of u = x W1^T + b1, d = gm W2 and the activation: dW1, db1, dW2 as
 * mdvit_mlp_rc_wgrad (same workspace, same fold) and the data gradient dx = du W1 as one partial per 256-wide hidden role, dx_parts [hidden / 256][M][C] -- the consumer adds
 * them (mdvit_layernorm_bwd2's dy2, or mdvit_sum_batch).  hidden in {256, 512}.  W1tp: the planes of W1^T ([2][C][hidden], mdvit_mlp_rc_dgrad's). */
int mdvit_mlp_rc_bwd(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, const void* W1tp, float* dx_parts, float* dW1, float* db1,
                     float* dW2, void* ws, size_t ws_bytes, int32_t M, int32_t C, int32_t hidden, float drop_p, uint32_t key1_0, uint32_t key1_1,
                     const uint32_t* drop_seed, int32_t accumulate, void* stream);
size_t mdvit_mlp_rc_wgrad_ws_bytes(int32_t M, int32_t C, int32_t hidden);
/* tuning hook (tools/mlp_rc_check.py): forward kernel variant -- 2: software-pipelined waves at 2 per SIMD, 3: plain waves at 3 per SIMD */
/* Launch ledger of mdvit_gemm_f32 (measurement only, off by default; bench.py's roofline line): mdvit_gemm_ledger(1) clears and starts it, (0) stops it; while on, every
 * mdvit_gemm_f32 call -- those of mdvit_block_fwd / _bwd included -- is counted under the kernel symbol it launches (mdvit_gemm_kernel_name, main kernel) with its algorithmic
 * flops 2 M N K and bytes 4 (M K + N K + M N (1 + extra outputs / epilogue operands)).  mdvit_gemm_ledger_read(i, ...) returns row i, MDVIT_E_SHAPE past the last row. */
int mdvit_gemm_ledger(int32_t enable);
int mdvit_gemm_ledger_read(int32_t index, char* name, int32_t cap, int64_t* launches, double* flop, double* bytes);
/* Launch sampler of mdvit_gemm_f32 (measurement only; bench.py's roofline line -- mdvit_amd has no reference counterpart: the reference has no native code): kernel begin / end
 * timestamps around a hashed 1-in-`stride` sample of the launches of `symbol` (NULL / "": of every kernel symbol), whoever issues them -- mdvit_block_fwd / _bwd included;
 * stride <= 0 switches it off.  mdvit_gemm_sampler_read(i, ...) waits for the sampled launches and returns symbol i's launches seen / timed and the summed duration [ms];
 * MDVIT_E_SHAPE past the last symbol.  One thread. */
int mdvit_gemm_sampler(const char* symbol, int32_t stride);
int mdvit_gemm_sampler_read(int32_t index, char* name, int32_t cap, int64_t* seen, int64_t* timed, double* ms);
int mdvit_mlp_rc_config(int32_t fwd_variant);
/* The arithmetic of the register-chained MLP kernels (mdvit_mlp_rc_* / mdvit_mlp_rc16_*): 2 (default) = bf16x3, the parity mode -- every operand as hi + lo bf16 planes,
 * three MFMAs per product; 1 = the bf16 speed mode of BASELINE configs[1] / [3] -- the hi plane alone, one MFMA per product and no lo split of the chained hidden operand
 * (~2^-9 per product instead of ~2^-17).  Process-wide, like the GEMM precision it follows (mdvit_amd.ops.set_gemm_precision sets both); operands and workspaces are the same. */
int mdvit_mlp_rc_planes(int32_t planes);
/* tuning hook: whether mdvit_block_bwd runs the C = 64 MLP backward as ONE kernel (mdvit_mlp_rc_bwd) -- 0 never, 1 always (default), 2 only when the call has no
 * weight-gradient stream (with one, the separate weight-gradient kernel overlaps the main stream's chain; measured, the one kernel still wins or ties).  Set it before mdvit_block_bwd_ws_bytes: the workspace layout follows it. */
int mdvit_block_config(int32_t mlp_bwd_fused);
/* tuning hook (tools/attn_time.py --apply-mode): how the attention backward's apply kernel (Ch = 8 / 16) orders its loads -- 0: each 32-token tile's rows in front of
 * the tile (default), 1: the MFMA operand rows one tile ahead at two waves per SIMD, 2: the same at one wave per SIMD.  Same arithmetic in every mode. */
int mdvit_factoratt_config(int32_t apply_mode, int32_t apply_tiles /* 32-token tiles per workgroup of the apply kernels; 0: the launcher's rule */);
int mdvit_mlp_rc_wgrad(const float* gm, const float* x, const void* W1p, const float* b1, const void* W2tp, float* dW1, float* db1, float* dW2,
                       void* ws, size_t ws_bytes, int32_t M, int32_t C, int32_t hidden, float drop_p, uint32_t key1_0, uint32_t key1_1,
                       const uint32_t* drop_seed, int32_t accumulate, void* stream);

/* ---- one entry per SerialBlock_adapt pass (csrc/block.hip) ----------------------------------------------------------------------------
 * SerialBlock_adapt.forward, mdvit.py:346-361 (and mpvit's SerialBlock, BASE: label == NULL):
 *     x1 = x + dwconv3x3(x) + b                                  ConvPosEnc                       mpvit.py:239-248
 *     x2 = x1 + DropPath(Dropout(proj(a * FactorAtt_ConvRelPosEnc(qkv(LN1(x1))))))                mdvit.py:281-313,353
 *     y  = x2 + DropPath(Dropout(fc2(Dropout(GELU(fc1(LN2(x2)))))))                               mpvit.py:71-78, mdvit.py:357-360
 * and its autograd backward, ENQUEUED from C: the same kernels in the same order as the operator-level entry points above (bit-identical
 * results), one host call per pass instead of ~25.  The caller owns three buffers: `save` (mdvit_block_save_bytes: everything the backward
 * needs, laid out by the library; kept between the passes), a forward workspace and a backward workspace (temporaries, split-K and
 * partial-sum scratch; the backward workspace must stay alive until the side stream has finished).
 * Weights: PyTorch layouts; `*_wt` = the transposed copies the bf16x3 data-gradient GEMMs read (mdvit_transpose_many), `fc*_p` = the bf16
 * planes of mlp_rc (mdvit_split_planes_many; NULL: the MLP runs as GEMMs).  precision: 0 fp32 MFMA, 1 bf16x3.  ln_groups: rows of the
 * [groups, C] LayerNorm parameters (MDViT_DSN banks on a domain-batched tensor; 1 otherwise).  label: [B, D] one-hot or NULL (no adapter).
 * Dropout: drop_p with one key pair per site (proj, fc1, fc2) as in the GEMM epilogues; rowscale1/2: DropPath scales [B] or NULL. */
typedef struct MdvitBlockDesc {
    int32_t B, H, W, C, heads, hidden, s3, s5, s7, ln_groups, D, da_hidden, precision;
    float eps, drop_p;
    uint32_t key_proj[2], key_fc1[2], key_fc2[2];
    const uint32_t* drop_seed;
    const float* rowscale1; const float* rowscale2;
    const float* label;
    const float *cpe_w, *cpe_b, *n1_g, *n1_b, *qkv_w, *qkv_b, *w3, *b3, *w5, *b5, *w7, *b7, *da_w1, *da_b1, *da_w2, *da_b2, *proj_w, *proj_b, *n2_g, *n2_b, *fc1_w, *fc1_b,
        *fc2_w, *fc2_b;
    const float *qkv_wt, *proj_wt, *fc1_wt, *fc2_wt;
    const void *fc1_p, *fc2_p, *fc2t_p, *fc1t_p;
    const void *qkv_p, *proj_p, *projt_p, *qkvt_p;      /* optional bf16 planes of Wqkv, Wproj, Wproj^T: C = 64 / 128 run qkv / proj / proj's data gradient on mdvit_linear_rc;
                                                * wider blocks run every product mdvit_gemm_ph_prefers accepts on the 256-wide plane kernel (fp32 activations split
                                                * while staged) when the weight's planes are given (forward: qkv_p, proj_p, fc1_p, fc2_p; data gradients: the transposed
                                                * planes fc2t_p, fc1t_p, projt_p, qkvt_p = planes of Wqkv^T [C, 3C]) */
    int32_t store_bf16;                        /* "mixed" mode (BASELINE configs[3]): the C = 128 MLP's saved hidden activation h and its gradient du -- the two
                                                * [tokens, hidden] tensors the block still moves, operands of weight-gradient GEMMs only -- are stored as bf16:
                                                * y and dx do not change by a bit, the fc1 / fc2 weight (and fc1 bias) gradients see bf16-rounded operands
                                                * (~2e-3 relative).  Ignored where the block has no such tensor (C = 64) or the 16-token MLP kernels do not run. */
    const float* a_pre;                        /* optional (round 5): the adapter's output a [B, C] for `label`, computed ahead by mdvit_da_fwd_many -- the forward then
                                                * launches no adapter kernel and the backward reads a from HERE (hand the same pointer to mdvit_block_bwd; the slot of a
                                                * in `save` stays unused).  NULL: the block computes a itself. */
    int32_t attn_kind;                         /* 0: SerialBlock_adapt (ConvPosEnc + factorised attention with ConvRelPosEnc, mdvit.py:346-361).
                                                * 1 (round 6): Block_adapt of the DeiT trunk of TransFuse_S_adapt (vision_transformer.py:191-211 with Attention_Sup,
                                                *    :125-169): x + proj(a * softmax(q k^T / sqrt(d)) v) of LN1(x), then x + Mlp(LN2(x)) -- no ConvPosEnc, no crpe windows
                                                *    (cpe_* / w3..b7 NULL, s3 = s5 = s7 = 0); the attention runs on mdvit_sdpa_mfma_fwd / _bwd, so H * W == 256 tokens,
                                                *    C == 64 heads, heads <= 6.  Same kernels in the same order as the operator-level path of mdvit_amd/transfuse.py. */
} MdvitBlockDesc;
/* Gradient outputs of the backward.  The sixteen "weight-class" outputs (cpe, qkv, crpe windows, proj, fc1, fc2) are overwritten
 * (accumulate == 0: fresh buffers) or added into (accumulate != 0: gradient buckets; the weight-gradient kernels then run on the side
 * stream).  The adapter gradients are always overwritten, the LayerNorm gradients too unless ln_accumulate.  dgrad_only: the data-gradient-only sweep of the merged two-sweep
 * step (multi_train_MDViT.py:198-207): no parameter gradient except the adapter's, NEGATED; aux_first: this block holds the first adapter of
 * the network -- that sweep ends here (dx is not produced). */
typedef struct MdvitBlockGrads {
    float *cpe_w, *cpe_b, *n1_g, *n1_b, *qkv_w, *qkv_b, *w3, *b3, *w5, *b5, *w7, *b7, *da_w1, *da_b1, *da_w2, *da_b2, *proj_w, *proj_b, *n2_g, *n2_b, *fc1_w, *fc1_b, *fc2_w,
        *fc2_b;
    int32_t accumulate, dgrad_only, aux_first;
    int32_t ln_accumulate;      /* != 0 (needs accumulate != 0 and a side stream): n1_g / n1_b / n2_g / n2_b and fc2_b are gradient buckets too -- their
                                   second-stage reductions ADD into them, on the side stream */
    float* e_out;               /* optional (round 5): [B, C]; the block writes e = a * dL/da of its adapter HERE and launches no adapter backward (da_w1 .. da_b2 are
                                   not touched) -- the caller forms every adapter's gradients in one go with mdvit_da_bwd_many.  NULL: the block runs mdvit_da_bwd itself. */
} MdvitBlockGrads;
/* main: the stream of the data-gradient chain; side: the stream of the weight-gradient kernels (NULL or == main: everything on main).
 * events: n_events HIP events owned by the caller (mdvit_event_create) for the main -> side ordering, used round robin from *next_event. */
typedef struct MdvitBlockStreams {
    void* main; void* side;
    void** events; int32_t n_events; int32_t* next_event;
} MdvitBlockStreams;
size_t mdvit_block_save_bytes(const MdvitBlockDesc* d);
size_t mdvit_block_fwd_ws_bytes(const MdvitBlockDesc* d);
int mdvit_block_fwd(const MdvitBlockDesc* d, const float* x, float* y, void* save, size_t save_bytes, void* ws, size_t ws_bytes, void* stream);
/* two backward workspaces: `ws` holds what only the main stream touches (free to reuse once the call has returned: stream order), `ws_side`
 * (*side_bytes) what the side stream's weight-gradient kernels read -- the caller keeps that one alive until the side stream is done */
size_t mdvit_block_bwd_ws_bytes(const MdvitBlockDesc* d, const MdvitBlockGrads* g, int32_t with_side_stream, size_t* side_bytes);
int mdvit_block_bwd(const MdvitBlockDesc* d, const MdvitBlockGrads* g, const MdvitBlockStreams* st, const float* x, const void* save, size_t save_bytes,
                    const float* dy, float* dx /* NULL: not wanted */, void* ws, size_t ws_bytes, void* ws_side, size_t ws_side_bytes);

/* y[m] (+)= dot(x[m,:K], w[:K]) + b   -- a 1-output-channel 1x1 conv (finalconv mdvit.py:589-591,
 * linear_out Decoders.py:311).  bwd: dx[m,k] = dy[m] w[k]; dw[k] = sum_m dy[m] x[m,k]; db = sum dy. */
int mdvit_rowdot_fwd(const float* x, int64_t ldx, const float* w, const float* b, float* y,
                     int32_t M, int32_t K, int32_t accumulate, void* stream);
int mdvit_rowdot_bwd(const float* x, int64_t ldx, const float* w, const float* dy, float* dx, int64_t lddx,
                     float* dw, float* db, void* ws /* n = K + 1 */, size_t ws_bytes, int32_t M, int32_t K, void* stream);

/* Masked upstream gradient + bias gradient in one pass over dY [M,N] (the backward of  y = drop(x W^T + b) * droppath,
 * mdvit.py:306-309,343-346):  masked[m][n] = A[m][n] * dropmask(m*N+n) * rowscale[m / rows_per_scale]  (optional) and
 * out[n] (+)= sum_m masked[m][n]  (optional).  The mask is re-derived from (key0,key1[,drop_seed]) exactly as the forward
 * GEMM epilogue drew it; the dgrad/wgrad GEMMs then read `masked` with no prologue of their own. */
int mdvit_colsum_f32(const float* A, int64_t lda, float* out, float* masked, void* ws /* n = N; unused if out == NULL */, size_t ws_bytes, int32_t M, int32_t N,
                     float drop_p, uint32_t key0, uint32_t key1, const float* rowscale, int32_t rows_per_scale,
                     int32_t accumulate, const uint32_t* drop_seed, void* stream);

/* ---- LayerNorm over C (nn.LayerNorm eps=1e-6, mdvit.py:327,342,498) -------------------------
 * groups: the M rows are `groups` equal consecutive row groups, group g normalised with parameter row g of
 * gamma/beta [groups, C] (the domain-specific norm banks of MDViT_DSN, mdvit.py:364-412,735-790, on a
 * domain-batched tensor); groups = 1 is the plain LayerNorm.  dgamma/dbeta are [groups, C]. */
int mdvit_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int32_t M, int32_t C, int32_t groups, float eps, void* stream);
/* add (optional, [M,C]): gradient arriving at x along the residual branch that forked off before the norm
 * (x + f(LN(x)), mdvit.py:353-360) -- dx = LN-backward(dy) + add in the same pass. */
int mdvit_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                        const float* add, float* dx, float* dgamma, float* dbeta, void* ws /* n = 2C */, size_t ws_bytes,
                        int32_t M, int32_t C, int32_t groups, void* stream);
/* The same backward with a SECOND output: dx_masked = dx * dropout mask(key0, key1) * rowscale[row / rows_per_scale] -- the masked upstream
 * gradient of the Linear whose output (after Dropout / DropPath, mdvit.py:310-311,353) is the LayerNorm's input, produced in the pass that
 * writes dx instead of a second pass over it (mdvit_colsum_f32's `masked` output, same arithmetic).  C in {64, 128, 320, 512}. */
int mdvit_layernorm_bwd_masked(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* add, float* dx,
                               float* dx_masked, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, int32_t M, int32_t C, int32_t groups,
                               float drop_p, uint32_t key0, uint32_t key1, const float* rowscale, int32_t rows_per_scale, const uint32_t* seed, void* stream);

/* ---- 3x3 convolutions on NHWC ------------------------------------------------------------------
 * dwconv3x3: depthwise, pad 1, stride 1|2, optional bias, optional "+ input" (ConvPosEnc,
 * mpvit.py:239-248; DWCPatchEmbed dwconv mdvit.py:90-97).  w is [C,1,3,3]. */
int mdvit_dwconv3x3_fwd(const float* x, const float* w, const float* bias, float* y,
                        int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t stride, int32_t add_input, void* stream);
int mdvit_dwconv3x3_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias,
                        void* ws /* n = 10C; unused if dw == NULL */, size_t ws_bytes,
                        int32_t B, int32_t Hi, int32_t Wi, int32_t C, int32_t stride, int32_t add_input,
                        int32_t accumulate /* dw, dbias += (gradient buckets) */, void* stream);
/* gconv2: Conv2d(2C, C, 3, groups=C, bias=False) applied to cat(skip, up) WITHOUT materialising the
 * concat (Decoders.py:30-38,198-199).  w is [C,2,3,3]; output channel g reads concat channels 2g, 2g+1. */
int mdvit_gconv2_3x3_fwd(const float* skip, const float* up, const float* w, float* y,
                         int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int mdvit_gconv2_3x3_bwd(const float* dy, const float* skip, const float* up, const float* w,
                         float* dskip, float* dup, float* dw, void* ws /* n = 18C; unused if dw == NULL */, size_t ws_bytes,
                         int32_t B, int32_t H, int32_t W, int32_t C, int32_t accumulate /* dw += */, void* stream);
/* Dense 3x3 (padding = dilation) as im2col + GEMM: col is [B*Ho*Wo, Cin*9], column order (cin,kh,kw) == weight.view(Cout,-1)
 * (stem.1 mpvit.py:104-111, bridge mdvit.py:557-564; dilation 6/12/18 at stride 1: the ASPP branches of the 'DeepLabV3'
 * peer heads, Utils/_deeplab.py:115-122). */
int mdvit_im2col3x3(const float* x, float* col, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, int32_t stride, int32_t dilation, void* stream);
int mdvit_col2im3x3(const float* dcol, float* dx, int32_t B, int32_t Hi, int32_t Wi, int32_t Cin, int32_t stride, int32_t dilation, void* stream);
/* y = x * dropmask / (1 - p), element-wise (nn.Dropout, Utils/_deeplab.py:155), the mask of the GEMM epilogues' counter hash:
 * the backward is the same call on the gradient with the same keys. */
int mdvit_dropout_f32(const float* x, float* y, int64_t n, float p, uint32_t key0, uint32_t key1, const uint32_t* drop_seed, void* stream);
/* stem.0: NCHW image [B,Cin,H,W] -> NHWC [B,H/2,W/2,Cout], 3x3 s2 p1, no bias (mdvit.py:509-517). */
int mdvit_stemconv_fwd(const float* img, const float* w, float* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream);
int mdvit_stemconv_wgrad(const float* img, const float* dy, float* dw, void* ws /* n = 27*Cout */, size_t ws_bytes,
                         int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t accumulate /* dw += */, void* stream);

/* ---- BatchNorm2d (train: batch stats, biased var; running stats momentum, unbiased var) + activation
 * on NHWC [M,C]  (mpvit.py:112-123, mdvit.py:99-122,559-563, Decoders.py:39-62,304-306).
 * ws: mdvit_bn_ws_bytes(M, C, groups) bytes of scratch.  Channel sums are reduced in a fixed order (per-block partial
 * rows, then a block-ordered double sum), so results are bitwise reproducible run to run.
 * groups: the M rows are `groups` equal consecutive row groups (one per domain batch, multi_train_MDViT.py:137-153
 * runs one forward per domain); statistics, normalisation and the backward sums are per group -- mean/rstd are
 * [groups, C] -- and the running statistics receive the groups' momentum updates in order, i.e. exactly what
 * `groups` consecutive forwards of M/groups rows each produce.  dgamma/dbeta are summed over the groups.
 * per_group_affine != 0 (MDViT_DSN's per-domain BatchNorm banks, mdvit.py:23-70,127-179): gamma/beta, the running
 * statistics and dgamma/dbeta are [groups, C], num_batches_tracked is [groups]; group g reads and updates row g only.
 * drop2d: nn.Dropout2d on (sample, channel) planes (Decoders.py:309,333). */
size_t mdvit_bn_ws_bytes(int32_t M, int32_t C, int32_t groups);
int mdvit_bn_stats(const float* y, void* ws, size_t ws_bytes, float* mean, float* rstd, float* running_mean, float* running_var,
                   int64_t* num_batches_tracked, int32_t M, int32_t C, int32_t groups, int32_t per_group_affine, float eps,
                   float momentum, void* stream);
int mdvit_bn_eval_prep(const float* running_mean, const float* running_var, float* mean, float* rstd, int32_t C, float eps, void* stream);
int mdvit_bn_apply(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta, float* z,
                   int32_t M, int32_t C, int32_t groups, int32_t per_group_affine, int32_t act, float drop2d_p, uint32_t key0,
                   uint32_t key1, const uint32_t* drop_seed, int32_t rows_per_sample, void* stream);
int mdvit_bn_bwd(const float* dz, const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                 float* dy, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, int32_t M, int32_t C, int32_t groups,
                 int32_t per_group_affine, int32_t act, int32_t training, float drop2d_p, uint32_t key0, uint32_t key1, const uint32_t* drop_seed, int32_t rows_per_sample,
                 void* stream);

/* BatchNorm2d -> activation -> Dropout2d -> 1-output 1x1 conv as ONE op: the tail of the peer heads (linear_fuse.1 -> ReLU -> Dropout2d(0.1) ->
 * linear_out, Decoders.py:304-311,333-336; the DeepLabV3 head's BN -> ReLU -> 1x1 conv, Utils/_deeplab.py).  The normalised [M, C] tensor between
 * the norm and the conv is never written: fwd reads y once; bwd (two passes over y and the row gradient g [M]) returns dy and, unless
 * dgamma / dbeta / dw are NULL (data gradient only), the four parameter gradients.  mean / rstd [C] come from mdvit_bn_stats (training) or
 * mdvit_bn_eval_prep (eval); one statistics group.  C in {256, 512, 1024}. */
size_t mdvit_bn_rowdot_ws_bytes(int32_t M, int32_t C);
int mdvit_bn_rowdot_fwd(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w, const float* b /* [1], optional */,
                        float* low, int32_t M, int32_t C, int32_t act, float drop2d_p, uint32_t key0, uint32_t key1, const uint32_t* drop_seed,
                        int32_t rows_per_sample, void* stream);
int mdvit_bn_rowdot_bwd(const float* g, const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w,
                        float* dy, float* dgamma, float* dbeta, float* dw, float* db /* optional */, void* ws, size_t ws_bytes, int32_t M, int32_t C, int32_t act,
                        int32_t training, float drop2d_p, uint32_t key0, uint32_t key1, const uint32_t* drop_seed, int32_t rows_per_sample, void* stream);

/* ---- bilinear resize, align_corners=False, NHWC (F.interpolate call sites mdvit.py:699,
 * Decoders.py:196,320-329,336) ------------------------------------------------------------------ */
int mdvit_upsample_fwd(const float* x, const float* base /* optional [B,Ho,Wo,C]: y = base + resize(x); may be y itself */, float* y,
                       int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C, void* stream);
size_t mdvit_upsample_bwd_ws_bytes(int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C);
/* adjoint, separable: dy [B,Ho,Wo,C] -> (width pass) ws [B,Ho,Wi,C] -> (height pass) dx [B,Hi,Wi,C] */
int mdvit_upsample_bwd(const float* dy, float* dx, void* ws, size_t ws_bytes, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                       int32_t C, void* stream);

/* y = base + sum_i resize(x_i) over n <= 3 NHWC sources of different sizes in ONE pass (the sum over the four encoder features in the peer heads'
 * fuse conv, Decoders.py:320-331: chained single-source calls move the [B,Ho,Wo,512] sum twice per source); base optional, may be y.  The
 * backward folds dy along W for all sources in one launch (workspace: sum_i B Ho Wi_i C floats) and along H per source.  C % 4 == 0.
 * xs / dxs / Hi / Wi are HOST arrays of n entries. */
int mdvit_upsample_multi_fwd(const float* const* xs, const int32_t* Hi, const int32_t* Wi, int32_t n, const float* base, float* y, int32_t B, int32_t Ho,
                             int32_t Wo, int32_t C, void* stream);
size_t mdvit_upsample_multi_bwd_ws_bytes(const int32_t* Wi, int32_t n, int32_t B, int32_t Ho, int32_t C);
int mdvit_upsample_multi_bwd(const float* dy, float* const* dxs, const int32_t* Hi, const int32_t* Wi, int32_t n, void* ws, size_t ws_bytes, int32_t B,
                             int32_t Ho, int32_t Wo, int32_t C, void* stream);

/* ---- Domain Adapter: a = softmax_heads(W2 relu(W1 label + b1) + b2), [B,C] (mdvit.py:272-276,301-303).
 * Backward takes e[b,c] = a[b,c] * dL/da[b,c] (what mdvit_factoratt_bwd emits -- it needs no division by a):
 * dz = scale * (e - a * sum_heads(e)).  ws: mdvit_da_ws_bytes(B, hid, C) bytes of scratch.  scale = -1 lets a
 * dgrad-only aux sweep pre-subtract its adapter gradient (mdvit_amd/train.py, multi_train_MDViT.py:198-207). */
int mdvit_da_fwd(const float* label, const float* W1, const float* b1, const float* W2, const float* b2, float* a,
                 int32_t B, int32_t D, int32_t hid, int32_t C, int32_t heads, void* stream);
/* The same for up to MDVIT_DA_MANY_MAX adapters and ONE label batch in one launch (round 5): an adapter's output depends on the labels and its own four tensors only, so a
 * model computes all of them at the top of its forward (mdvit_amd.ops.da_precomputed) and hands each block its slice through MdvitBlockDesc.a_pre -- instead of one ~13 us
 * launch inside every block on the single-stream forward.  a[i]: [B, C[i]] fp32.  Same arithmetic as mdvit_da_fwd, bit for bit. */
#define MDVIT_DA_MANY_MAX 32
typedef struct MdvitDaMany {
    int32_t n;
    int32_t hid[MDVIT_DA_MANY_MAX], C[MDVIT_DA_MANY_MAX], heads[MDVIT_DA_MANY_MAX];
    const float* W1[MDVIT_DA_MANY_MAX]; const float* b1[MDVIT_DA_MANY_MAX]; const float* W2[MDVIT_DA_MANY_MAX]; const float* b2[MDVIT_DA_MANY_MAX];
    float* a[MDVIT_DA_MANY_MAX];
} MdvitDaMany;
int mdvit_da_fwd_many(const MdvitDaMany* m, const float* label, int32_t B, int32_t D, void* stream);
/* The backward of all of them in two launches per sweep: e[i] [B, C[i]] = a * dL/da as the attention backward of adapter i's block hands it out
 * (MdvitBlockGrads.e_out; mdvit_factoratt_bwd's e), m->a[i] the forward's outputs; the four gradients of every adapter with e[i] != NULL are OVERWRITTEN with
 * scale * (the gradient) -- scale = -1 in the data-gradient-only sweep, as mdvit_da_bwd.  e[i] == NULL: adapter i is skipped.  ws: mdvit_da_many_ws_bytes(m, B).
 * Same arithmetic as mdvit_da_bwd per adapter, bit for bit. */
typedef struct MdvitDaManyGrads {
    const float* e[MDVIT_DA_MANY_MAX];
    float* dW1[MDVIT_DA_MANY_MAX]; float* db1[MDVIT_DA_MANY_MAX]; float* dW2[MDVIT_DA_MANY_MAX]; float* db2[MDVIT_DA_MANY_MAX];
} MdvitDaManyGrads;
size_t mdvit_da_many_ws_bytes(const MdvitDaMany* m, int32_t B);
int mdvit_da_bwd_many(const MdvitDaMany* m, const MdvitDaManyGrads* g, const float* label, float scale, void* ws, size_t ws_bytes, int32_t B, int32_t D, void* stream);
size_t mdvit_da_ws_bytes(int32_t B, int32_t hid, int32_t C);
int mdvit_da_bwd(const float* label, const float* W1, const float* b1, const float* W2, const float* b2, const float* a,
                 const float* e, float scale, float* dW1, float* db1, float* dW2, float* db2, void* ws, size_t ws_bytes,
                 int32_t B, int32_t D, int32_t hid, int32_t C, int32_t heads, void* stream);

/* ---- factorized attention core (mdvit.py:293-304, mpvit.py:296-318) -------------------------
 * qkv: [B,N,3C] as produced by the qkv Linear (q | k | v, channel = head*Ch + ch).
 * U[b,n,c]   = dwconv_win(v)[n,c] + bias[c]                       (saved for backward)
 * out[b,n,c] = a[b,c] * ( Ch^-0.5 * sum_j q[n,head,j] M[b,head,j,ch] + q[n,c] * U[b,n,c] )
 * with M = softmax_over_tokens(k)^T v.  crpe weights: [s3*Ch,1,3,3], [s5*Ch,1,5,5], [s7*Ch,1,7,7] (+bias).
 * a == NULL: no domain adapter (mpvit.py:347-373).  kmax/ksum [B,C] and Mmat [B,C,Ch] are saved for backward.
 * Backward returns dqkv, the crpe gradients (all six may be NULL: dgrad only) and e = a * dL/da (NULL when a is NULL).  dqkv == NULL: e alone (one pass over dout and out) --
 * what the data-gradient-only sweep needs at the first adapter of the network. */
size_t mdvit_factoratt_ws_bytes(int32_t B, int32_t N, int32_t C, int32_t heads);
int mdvit_factoratt_fwd(const float* qkv, const float* w3, const float* b3, const float* w5, const float* b5,
                        const float* w7, const float* b7, const float* a, float* out, float* U, float* kmax, float* ksum, float* Mmat,
                        void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                        int32_t s3, int32_t s5, int32_t s7, void* stream);
int mdvit_factoratt_bwd(const float* dout, const float* qkv, const float* out, const float* U,
                        const float* w3, const float* b3, const float* w5, const float* b5, const float* w7, const float* b7,
                        const float* a, const float* kmax, const float* ksum, const float* Mmat,
                        float* dqkv, float* e, float* dw3, float* db3, float* dw5, float* db5, float* dw7, float* db7,
                        void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                        int32_t s3, int32_t s5, int32_t s7, void* stream);
/* The six window-weight gradients of the preceding mdvit_factoratt_bwd call (which was given NULL for them): they read
 * only dU (left in the SAME ws) and v, so the caller may issue them on another stream -- ordered after that backward --
 * and let them overlap the rest of the data-gradient chain.  accumulate != 0: add into dw/db (gradient buckets). */
int mdvit_factoratt_wgrad(const float* qkv, void* ws, size_t ws_bytes, float* dw3, float* db3, float* dw5, float* db5,
                          float* dw7, float* db7, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                          int32_t s3, int32_t s5, int32_t s7, int32_t accumulate, void* stream);

/* ---- step losses on logits (multi_train_MDViT.py:147-169, Utils/losses.py:8-16, nn.BCELoss) ---
 * losses[0] = BCE(s(out),y)+Dice(s(out),y); [1] = same for aux; [2] = Dice(s(aux), s(out)).  aux may be NULL.
 * sums: 16 doubles of scratch kept for backward.  g: 3 upstream gradients (device memory). */
int mdvit_seg_losses_fwd(const float* out, const float* aux, const float* label, double* sums, float* losses, int64_t n, void* stream);
/* The same in two halves, for data parallelism: nn.DataParallel gathers the replicas' outputs and takes BCE / Dice over the
 * GLOBAL batch (multi_train_MDViT.py:72-74,147-153).  Each rank runs _sums on its n local elements, the 16 doubles are
 * all-reduced (sum), _final turns them into the global losses with n_total = n * world.  In _bwd, n stays the LOCAL count
 * and dice_gain = world: local logit gradients are world * dL_global/dlogits, which the gradient AVERAGE over ranks turns
 * into dL_global/dtheta.  (Single process: dice_gain = 1, n_total = n -- identical to mdvit_seg_losses_fwd.) */
int mdvit_seg_losses_sums(const float* out, const float* aux, const float* label, double* sums, int64_t n, void* stream);
int mdvit_seg_losses_final(const double* sums, float* losses, int64_t n_total, int32_t has_aux, void* stream);
int mdvit_seg_losses_bwd(const float* out, const float* aux, const float* label, const double* sums, const float* g,
                         float* dout, float* daux, int64_t n, float dice_gain, void* stream);
/* the same with the three upstream gradients (d total / d loss_out, d loss_aux, d loss_kt: 0-dim device tensors as autograd hands them) as separate pointers, NULL = that loss
 * takes no part in this sweep (the aux sweep of multi_train_MDViT.py:198-207 back-propagates loss_aux + kt only): no host-side stacking of the three into one buffer */
int mdvit_seg_losses_bwd3(const float* out, const float* aux, const float* label, const double* sums, const float* g0, const float* g1, const float* g2,
                          float* dout, float* daux, int64_t n, float dice_gain, void* stream);
/* The G domain batches of ONE domain-batched forward (multi_train_MDViT.py:137-169 computes the three losses per domain and adds them up): out / aux / label hold G consecutive
 * batches of n elements; sums [G][16]; losses [3] = the per-batch losses added in batch order (fp32, as the step's own additions were), per_group [G][3] optional.  Data parallel:
 * _groups_sums on the local elements, all-reduce the G x 16 doubles, _groups_final with n_total = n * world; in _groups_bwd n stays the local count per batch (see above). */
int mdvit_seg_losses_groups_sums(const float* out, const float* aux, const float* label, double* sums, int64_t n, int32_t G, void* stream);
int mdvit_seg_losses_groups_final(const double* sums, float* losses, float* per_group, int64_t n_total, int32_t has_aux, int32_t G, void* stream);
int mdvit_seg_losses_groups_bwd(const float* out, const float* aux, const float* label, const double* sums, const float* g0, const float* g1, const float* g2,
                                float* dout, float* daux, int64_t n, int32_t G, float dice_gain, void* stream);

/* ---- on-device metrics and input pipeline --------------------------------------------------------------
 * Dice / IoU of the thresholded outputs, multi_train_MDViT.py:172-179,275-288 (medpy.metric.binary dc / jc, v0.4.0:
 * dc = 2|A&Y| / (|A|+|Y|), jc = |A&Y| / |A|Y|) with A = sigmoid(logits) > 0.5, Y = label != 0 -- computed where the logits
 * live instead of `.cpu().numpy()` per domain.  counts: 8 uint64 of scratch (exact integer counts: |A&Y|, |A|, |Y|,
 * |Aaux&Y|, |Aaux|); metrics: {dice, iou, aux dice, aux iou} (0 where medpy divides 0/0).  aux may be NULL. */
int mdvit_seg_metrics(const float* out, const float* aux, const float* label, uint64_t* counts, float* metrics, int64_t n, void* stream);
/* uint8 HWC image [B,H,W,3] -> fp32 CHW [B,3,H,W]:  ((float)(u8 / 255.0) - mean[c]) / std[c]  with the ImageNet mean / std
 * (create_dataset.py:25-26 norm01, :143-144,165-172 permute + transforms.Normalize), bit-exact with that sequence. */
int mdvit_image_normalize_u8(const uint8_t* img_nhwc, float* out_nchw, int32_t B, int32_t H, int32_t W, void* stream);

/* ---- AdamW over all parameters in one launch (optim.AdamW, multi_train_MDViT.py:91-93) --------------------
 * table_dev: device array [n_tensors][5] of int64 {param, grad, exp_avg, exp_avg_sq (pointers), numel}; every row gets blocks_per_tensor
 * workgroups, so a caller with very unequal tensors hands in one row per CHUNK of a tensor (mdvit_amd/optim.py: 32768 elements).  step_dev[0] (float
 * step count) is incremented first, lr_dev[0] is the current learning rate -- both in device memory so that a captured
 * HIP graph replays the update unchanged while the host moves the schedule.  zero_grad != 0 clears each gradient after use.
 * Math: torch.optim.AdamW (decoupled decay, bias-corrected, amsgrad off). */
int mdvit_adamw_step(const void* table_dev, int32_t n_tensors, int32_t blocks_per_tensor, const float* lr_dev, float* step_dev,
                     float beta1, float beta2, float eps, float weight_decay, int32_t zero_grad, void* stream);

/* ---- TransFuse_S_adapt path (BASELINE configs[4]; csrc/transfuse.hip) ------------------------------------------------------------
 * What Models/Hybrid_models/TransFuseFolder/{TransFuse,vision_transformer,DeiT}.py and multi_train_TransFuse.py need beyond the MDViT
 * kernels above (dense 3x3 / 1x1 convolutions, BatchNorm, LayerNorm, Linear / MLP GEMMs and the Domain Adapter are shared).  NHWC fp32. */
/* ResNet conv1: KxK stride 2 pad K/2 on the NCHW image -> NHWC (torchvision resnet conv1; TransFuse.py:243).  ksize = 7, in_chans = 3.
 * wgrad ws: mdvit_partials_ws_bytes(147 * Cout) */
int mdvit_imgconv_fwd(const float* img, const float* w, float* y, int32_t B, int32_t H, int32_t W, int32_t in_chans, int32_t Cout, int32_t ksize, void* stream);
int mdvit_imgconv_wgrad(const float* img, const float* dy, float* dw, void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t in_chans,
                        int32_t Cout, int32_t ksize, int32_t accumulate, void* stream);
/* the same conv's im2col: col [B Ho Wo, ldc] with columns (ci, kh, kw) zero-padded to ldc (>= 147, % 4 == 0): stem conv and weight gradient as GEMMs */
int mdvit_imgconv_im2col(const float* img, float* col, int32_t B, int32_t H, int32_t W, int32_t in_chans, int32_t ksize, int32_t ldc, void* stream);
/* nn.MaxPool2d(3, 2, 1) (TransFuse.py:246); idx: uint8 winning tap per output element */
int mdvit_maxpool3x3s2_fwd(const float* x, float* y, void* idx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
int mdvit_maxpool3x3s2_bwd(const float* dy, const void* idx, float* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
/* bilinear resize with align_corners=True (nn.Upsample in Up, TransFuse.py:528; F.interpolate of the heads, :267-269) */
int mdvit_resize_ac_fwd(const float* x, float* y, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C, void* stream);
int mdvit_resize_ac_bwd(const float* dy, float* dx, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, int32_t C, void* stream);
/* element-wise, n elements.  mode 0: y = relu(a + b) (b optional)   1: y = a * b   2: y = a * (b > 0)   3: y = a + b
 * (BasicBlock / DoubleConv / Attention_block add+ReLU: TransFuse.py:598,573; W_g*W_x: :57; their backward) */
int mdvit_ew(const float* a, const float* b, float* y, int64_t n, int32_t mode, void* stream);
/* y = (a + b) + c: the sum of the three gradients of a tensor with three consumers (autograd's accumulation at a fan-out, mdvit.py:640-700: every encoder stage's
 * output feeds the next stage, the decoder's skip path and the peer heads) in one pass. */
int mdvit_add3(const float* a, const float* b, const float* c, float* y, int64_t n, void* stream);
/* ... where the third gradient arrives as G <= 8 equal consecutive parts (one per peer head: multi_train_MDViT.py:137-153 runs one forward per domain, the domain-batched
 * forward hands each head its batch group): y = (a + b) + concat(parts), the concatenation never written; b may be NULL. */
int mdvit_add_parts(const float* a, const float* b, const void* const* parts, int32_t G, int64_t part_elems, float* y, void* stream);
/* y[b, r] = x[b, r] + pe[r] (DeiT_adapt.forward x + pos_embed, DeiT.py:63-65); out[r] = sum_b g[b, r] (its gradient) */
int mdvit_add_bcast(const float* x, const float* pe, float* y, int32_t B, int64_t R, void* stream);
int mdvit_sum_batch(const float* g, float* out, int32_t B, int64_t R, void* stream);
/* y = sigmoid(s) * x on x [B, P, C]; mode 0: s [B, P] (spatial attention / Attention_block psi: TransFuse.py:63,576), mode 1: s [B, C]
 * (SE channel attention: :71).  bwd: dx = g sigmoid(s), ds = sigmoid'(s) * sum (g x) over the broadcast axis */
int mdvit_gate_fwd(const float* x, const float* s, float* y, int32_t B, int64_t P, int32_t C, int32_t mode, void* stream);
size_t mdvit_gate_bwd_ws_bytes(int32_t B, int64_t P, int32_t C, int32_t mode);
int mdvit_gate_bwd(const float* g, const float* x, const float* s, float* dx, float* ds, void* ws, size_t ws_bytes, int32_t B, int64_t P, int32_t C, int32_t mode,
                   void* stream);
/* ChannelPool (TransFuse.py:20-22): x [M, C] -> y [M, 2] = (max_c, mean_c); idx [M] int32 argmax */
int mdvit_chanpool_fwd(const float* x, float* y, int32_t* idx, int64_t M, int32_t C, void* stream);
int mdvit_chanpool_bwd(const float* dy, const int32_t* idx, float* dx, int64_t M, int32_t C, void* stream);
/* BiFusion_block.spatial: 7x7 conv, 2 -> 1 channels, pad 3, no bias (TransFuse.py:37): x [B,H,W,2], w [1,2,7,7] -> y [B,H,W] */
int mdvit_conv7x7_2to1_fwd(const float* x, const float* w, float* y, int32_t B, int32_t H, int32_t W, void* stream);
int mdvit_conv7x7_2to1_bwd(const float* dy, const float* x, const float* w, float* dx /* optional */, float* dw /* optional, [98] */, int32_t B, int32_t H,
                           int32_t W, void* stream);
/* BatchNorm2d(1) (spatial.bn, psi.1): x [M] -> y [M]; `groups` equal consecutive slices keep their own statistics (the domain-batched forward);
 * stat [groups][2] = (mean, rstd) kept for the backward; dgamma_dbeta [2] */
int mdvit_bn1_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* num_batches_tracked, float* y,
                  float* stat, int64_t M, int32_t groups, int32_t training, float eps, float momentum, void* stream);
int mdvit_bn1_bwd(const float* g, const float* x, const float* gamma, const float* stat, float* dx, float* dgamma_dbeta, int64_t M, int32_t groups, int32_t training,
                  void* stream);
/* every second pixel (the 1x1 stride-2 shortcut convs of ResNet layer2.0 / layer3.0); backward = 1: scatter src [B,H/2,W/2,C] into dst [B,H,W,C] */
int mdvit_subsample2(const float* src, float* dst, int32_t B, int32_t H, int32_t W, int32_t C, int32_t backward, void* stream);
/* PatchEmbed's gather (vision_transformer.py:233-240): NCHW image -> [B (H/p) (W/p), Cin p p] rows in (c, ky, kx) order (then a Linear) */
int mdvit_patchify(const float* img, float* out, int32_t B, int32_t in_chans, int32_t H, int32_t W, int32_t patch, void* stream);
/* nn.Dropout2d on x [B, P, C]: one keep decision per (sample, channel); the backward is the same call on the gradient */
int mdvit_dropout2d(const float* x, float* y, int32_t B, int64_t P, int32_t C, float p, uint32_t key0, uint32_t key1, const uint32_t* drop_seed, void* stream);
/* Attention_Sup core (vision_transformer.py:148-169): out[b,n,(h d)] = a[b,(h d)] * (softmax(q k^T / sqrt(d)) v)[b,h,n,d] from qkv [B,N,3C]
 * (q|k|v, channel = head * 64 + d); a [B, C] = the Domain Adapter's head-softmax (mdvit_da_fwd) or NULL.  P [B,heads,N,N] (softmax
 * probabilities) is kept for the backward; dS: scratch of the same size.  e [B, C] = a * dL/da = sum_n g out (input of mdvit_da_bwd).
 * head dimension 64, N <= 256. */
int mdvit_sdpa_fwd(const float* qkv, const float* a, float* out, float* P, int32_t B, int32_t N, int32_t C, int32_t heads, void* stream);
int mdvit_sdpa_bwd(const float* g, const float* qkv, const float* P, const float* out, const float* a, float* dqkv, float* e, float* dS, int32_t B, int32_t N,
                   int32_t C, int32_t heads, void* stream);
/* The same operator on the fp32 matrix cores for N == 256 tokens (the DeiT trunk's 16 x 16 grid), head dimension 64, <= 6 heads: no [N, N]
 * tensor in HBM -- the forward keeps lse [B, heads, N] (row log-sum-exp), the backward recomputes the probabilities.
 * delta: scratch [2][B, heads, N] floats. */
int mdvit_sdpa_mfma_fwd(const float* qkv, const float* a, float* out, float* lse, int32_t B, int32_t N, int32_t C, int32_t heads, void* stream);
int mdvit_sdpa_mfma_bwd(const float* g, const float* qkv, const float* lse, const float* out, const float* a, float* dqkv, float* e, float* delta,
                        int32_t B, int32_t N, int32_t C, int32_t heads, void* stream);
/* structure_loss (multi_train_TransFuse.py:29-38): weit = 1 + 5 |avg_pool2d(mask, 31, 1, 15) - mask| (tmp: scratch [B,H,W]);
 * loss = mean_b [ sum(weit bce_with_logits) / sum(weit) + 1 - (I + 1) / (U - I + 1) ], I = sum(sigmoid(pred) mask weit), U = sum((sigmoid(pred) + mask) weit);
 * sums [B][4] double kept for the backward; gscale [1] = upstream gradient of the scalar loss */
int mdvit_structure_weight(const float* mask, float* tmp, float* weit, int32_t B, int32_t H, int32_t W, void* stream);
int mdvit_structure_loss_fwd(const float* pred, const float* mask, const float* weit, double* sums, float* loss, int32_t B, int64_t HW, void* stream);
int mdvit_structure_loss_bwd(const float* pred, const float* mask, const float* weit, const double* sums, const float* gscale, float* dpred, int32_t B, int64_t HW,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDVIT_HIP_H */
