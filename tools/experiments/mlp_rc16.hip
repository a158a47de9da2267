// EXPERIMENT (round 3, measured, not part of the library build): the forward kernel of csrc/mlp_rc.hip on 16x16x32 MFMA tiles -- a wave owns 16
// tokens, half the registers per wave, C = 64 at four waves per SIMD and C = 128 at two.  Correct (1.1e-6 .. 1.7e-6 of the operator path at
// C = 64 and C = 128, tools/mlp_rc16_check.py at commit "rc16 experiment") and NOT faster: C = 64 at 524288 tokens 412 us against 347 us for
// the 32x32 kernel at three waves per SIMD (twice the LDS operand bytes per hidden element buy nothing: the activation VALU, not occupancy,
// is the limit); C = 128 at 131072 tokens 350 us against ~410 us for the two GEMMs it would replace -- but dropping `h` at C = 128 also needs
// the recomputing weight-gradient kernel, whose four K = 128 products cost more matrix time than the two weight-gradient GEMMs save in
// bytes.  The text below is the section as it stood inside mlp_rc.hip (it uses that file's helpers) plus the host dispatch.
#if 0

// ==============================================================================================================================
// The same three kernels on 16x16x32 MFMA tiles ("rc16"): a wave owns 16 tokens (forward, data gradient) or 16 hidden units (weight
// gradients).  Half the registers per wave -> twice the waves per SIMD for the VALU-bound activation work (the 32x32 forms run 2-3 waves per
// SIMD at ~1 VALU instruction per 5 cycles per SIMD), and C = 128 fits: the C = 128 stages' MLP (mpvit.py:71-78, hidden 1024) gets the
// no-[tokens, hidden]-tensor treatment too.
//   v_mfma_f32_16x16x32_bf16: A lane (i = l & 15, k = 8 (l >> 4) ..+7), B lane (j = l & 15, same k), D lane j, registers i = 4 (l >> 4) + r.
// Product 1 as D[hidden][token] leaves a lane with the hidden units 4g .. 4g+3 of each 16-row tile (g = l >> 4); the second product
// contracts over a 32-wide hidden step with k slot (g, i) <-> hidden (i < 4 ? 4g + i : 16 + 4g + i - 4): the weight operand is READ in that
// order (two ds_read_b64 per fragment), the chained operand needs no lane exchange at all.
// ==============================================================================================================================
typedef float rc_f32x4 __attribute__((ext_vector_type(4)));

namespace {

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

#define RC16_MFMA3(acc, ah, al, bh, bl)                                         \
    do {                                                                        \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);    \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);    \
    } while (0)

// x / gm operand fragments of a wave's 16 tokens: lane (token l & 15, g = l >> 4) holds c = 32 ks + 8 g ..+7 for every k step
template <int C>
__device__ __forceinline__ void rc16_load_rows(const float* __restrict__ src, int row, int M, int g, rc_bf16x8 (&hi)[C / 32], rc_bf16x8 (&lo)[C / 32]) {
    const float* p = src + (long)min(row, M - 1) * C + 8 * g;
    float4 a[C / 32], b[C / 32];
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
        a[ks] = *reinterpret_cast<const float4*>(p + 32 * ks);
        b[ks] = *reinterpret_cast<const float4*>(p + 32 * ks + 4);
    }
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
        const float v[8] = {a[ks].x, a[ks].y, a[ks].z, a[ks].w, b[ks].x, b[ks].y, b[ks].z, b[ks].w};
        rc_u4 h, l;
        rc_split8(v, h, l);
        hi[ks] = __builtin_bit_cast(rc_bf16x8, h);
        lo[ks] = __builtin_bit_cast(rc_bf16x8, l);
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------------------------
template <int C, int NW, int OCC, bool DROP>
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
    rc16_load_rows<C>(p.x, row, p.M, g, xh, xl);
    rc_f32x4 yacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) yacc[ct] = rc_f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int t = 0; t < n; ++t) {
        RC_WAIT_VM(PPW);
        __builtin_amdgcn_s_barrier();
        issue_group(t + 2);
        const int slot = t % 3;
        const char* w1h = sW1 + (slot * 2) * T1; const char* w1l = w1h + T1;
        const char* w2h = sW2 + (slot * 2) * T2; const char* w2l = w2h + T2;
        float hv[8];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const int hd = t * 32 + 16 * ht + 4 * g;              // this lane's four hidden units of the tile
            const rc_f4 b4 = *reinterpret_cast<const rc_f4*>(sB1 + hd);
            rc_f32x4 u = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const rc_bf16x8 ah = rc16_frag<RB1>(w1h, 16 * ht + c16, 4 * ks + g), al = rc16_frag<RB1>(w1l, 16 * ht + c16, 4 * ks + g);
                RC16_MFMA3(u, ah, al, xh[ks], xl[ks]);
            }
            float4 v = make_float4(rc_gelu(u[0]), rc_gelu(u[1]), rc_gelu(u[2]), rc_gelu(u[3]));
            if (DROP) {
                const float4 ds = mdvit_drop_scale4(k1a, k1b, (uint32_t)((long)row * p.Hd + hd), p.thresh, p.inv_keep);
                v.x *= ds.x; v.y *= ds.y; v.z *= ds.z; v.w *= ds.w;
            }
            hv[4 * ht + 0] = v.x; hv[4 * ht + 1] = v.y; hv[4 * ht + 2] = v.z; hv[4 * ht + 3] = v.w;
        }
        rc_u4 hh4, hl4;
        rc_split8(hv, hh4, hl4);
        const rc_bf16x8 hh = __builtin_bit_cast(rc_bf16x8, hh4), hl = __builtin_bit_cast(rc_bf16x8, hl4);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const rc_bf16x8 ah = rc16_frag_perm(w2h, 16 * ct + c16, g), al = rc16_frag_perm(w2l, 16 * ct + c16, g);
            RC16_MFMA3(yacc[ct], ah, al, hh, hl);
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

}  // namespace


// ---- host dispatch (inside mdvit_mlp_rc_fwd) ----
    if (C == 128 || g_rc_fwd_variant == 16) {
        // 16-token waves on 16x16x32 tiles: C = 64 at four waves per SIMD (two 8-wave workgroups per CU), C = 128 at two
        const int smem16 = 3 * 2 * (32 * C * 2) + 3 * 2 * (C * 64) + Hd * 4;
        static bool f0[64] = {false}, f1[64] = {false}, f2[64] = {false}, f3[64] = {false};
        int rc16 = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc16_fwd_kernel<64, 8, 4, false>), 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, f0);
        if (rc16 == MDVIT_OK) rc16 = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc16_fwd_kernel<64, 8, 4, true>), 3 * 2 * (32 * 128) + 3 * 2 * (64 * 64) + 4096 * 4, f1);
        if (rc16 == MDVIT_OK) rc16 = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc16_fwd_kernel<128, 8, 2, false>), 3 * 2 * (32 * 256) + 3 * 2 * (128 * 64) + 4096 * 4, f2);
        if (rc16 == MDVIT_OK) rc16 = rc_set_lds(reinterpret_cast<const void*>(&mlp_rc16_fwd_kernel<128, 8, 2, true>), 3 * 2 * (32 * 256) + 3 * 2 * (128 * 64) + 4096 * 4, f3);
        if (rc16 != MDVIT_OK) return rc16;
        const dim3 grid(cdiv(M, 8 * 16)), block(512);
        if (C == 64) {
            if (a.drop) hipLaunchKernelGGL((mlp_rc16_fwd_kernel<64, 8, 4, true>), grid, block, smem16, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((mlp_rc16_fwd_kernel<64, 8, 4, false>), grid, block, smem16, (hipStream_t)stream, a);
        } else {
            if (a.drop) hipLaunchKernelGGL((mlp_rc16_fwd_kernel<128, 8, 2, true>), grid, block, smem16, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((mlp_rc16_fwd_kernel<128, 8, 2, false>), grid, block, smem16, (hipStream_t)stream, a);
        }
        MDVIT_LAUNCH_CHECK();
        return MDVIT_OK;
    }
#endif
