// Factorized attention core with convolutional relative position encoding and the Domain Adapter
// (FactorAtt_ConvRelPosEnc_Sup.forward, mdvit.py:293-304; ConvRelPosEnc.forward, mpvit.py:296-318),
// forward and backward.  There is no N x N score matrix: softmax runs over the
// TOKEN axis of K per (batch, channel) column, M = softmax(K)^T V is Ch x Ch per head.
//
//   fwd  A: per (token tile, head chunk, batch): tile column max / exp-sum / partial K^T V    -> ws
//        B: combine tiles (rescale by exp(m_t - m)) -> M [B,C,Ch], column stats kmax/ksum [B,C]
//        C: out = a * (Ch^-0.5 * q.M + q * (dwconv_{3|5|7}(v) + bias))
//   bwd  1: token-axis reductions da, dM, d(crpe weights)      2: t = sum dM*M     3: dq,dk,dv
#include "common.h"

namespace {

constexpr int FA_T = 64;      // tokens per tile in fwd pass A
constexpr int FA_TCHUNK = 8;  // tokens exchanged through LDS per step in bwd pass 1

struct FaGeom {
    int B, H, W, N, C, heads, Ch, s3, s5, s7;
    float scale;
};

// window radius and weight pointer of channel c
struct CrpeW { const float* w3; const float* b3; const float* w5; const float* b5; const float* w7; const float* b7; };

__device__ __forceinline__ int crpe_radius(const FaGeom& g, int c) {
    const int head = c / g.Ch;
    return head < g.s3 ? 1 : (head < g.s3 + g.s5 ? 2 : 3);
}
__device__ __forceinline__ const float* crpe_wptr(const FaGeom& g, const CrpeW& cw, int c, int r) {
    if (r == 1) return cw.w3 + (long)c * 9;
    if (r == 2) return cw.w5 + (long)(c - g.s3 * g.Ch) * 25;
    return cw.w7 + (long)(c - (g.s3 + g.s5) * g.Ch) * 49;
}
__device__ __forceinline__ float crpe_bias(const FaGeom& g, const CrpeW& cw, int c, int r) {
    if (r == 1) return cw.b3[c];
    if (r == 2) return cw.b5[c - g.s3 * g.Ch];
    return cw.b7[c - (g.s3 + g.s5) * g.Ch];
}

// ---- fwd A --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fa_kv_partial_kernel(const float* __restrict__ qkv, float* __restrict__ ws_m,
                                                            float* __restrict__ ws_s, float* __restrict__ ws_P, FaGeom g, int CW, int NT) {
    extern __shared__ float sm[];          // ks[FA_T][CW], vs[FA_T][CW]
    float* ks = sm;
    float* vs = sm + FA_T * CW;
    const int tile = blockIdx.x, chunk = blockIdx.y, b = blockIdx.z;
    const int c0 = chunk * CW, n0 = tile * FA_T, nt = min(FA_T, g.N - n0);
    const int C3 = 3 * g.C;
    for (int i = threadIdx.x; i < nt * CW; i += blockDim.x) {
        const int n = i / CW, c = i % CW;
        const float* row = qkv + ((long)b * g.N + n0 + n) * C3;
        ks[n * CW + c] = row[g.C + c0 + c];
        vs[n * CW + c] = row[2 * g.C + c0 + c];
    }
    __syncthreads();
    if (threadIdx.x < CW) {
        const int c = threadIdx.x;
        float m = -INFINITY;
        for (int n = 0; n < nt; ++n) m = fmaxf(m, ks[n * CW + c]);
        float s = 0.f;
        for (int n = 0; n < nt; ++n) { const float e = expf(ks[n * CW + c] - m); ks[n * CW + c] = e; s += e; }
        const long o = ((long)b * NT + tile) * g.C + c0 + c;
        ws_m[o] = m; ws_s[o] = s;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < CW * g.Ch; o += blockDim.x) {
        const int c = o / g.Ch, e = o % g.Ch, hb = (c / g.Ch) * g.Ch;
        float acc = 0.f;
        for (int n = 0; n < nt; ++n) acc = fmaf(ks[n * CW + c], vs[n * CW + hb + e], acc);
        ws_P[(((long)b * NT + tile) * g.C + c0 + c) * g.Ch + e] = acc;
    }
}

// ---- fwd B --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fa_kv_combine_kernel(const float* __restrict__ ws_m, const float* __restrict__ ws_s,
                                                            const float* __restrict__ ws_P, float* __restrict__ kmax, float* __restrict__ ksum,
                                                            float* __restrict__ Mmat, FaGeom g, int NT) {
    const int b = blockIdx.y;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= g.C * g.Ch) return;
    const int c = o / g.Ch, e = o % g.Ch;
    float m = -INFINITY;
    for (int t = 0; t < NT; ++t) m = fmaxf(m, ws_m[((long)b * NT + t) * g.C + c]);
    float s = 0.f, acc = 0.f;
    for (int t = 0; t < NT; ++t) {
        const long i = ((long)b * NT + t) * g.C + c;
        const float f = expf(ws_m[i] - m);
        s = fmaf(ws_s[i], f, s);
        acc = fmaf(ws_P[i * g.Ch + e], f, acc);
    }
    Mmat[((long)b * g.C + c) * g.Ch + e] = acc / s;
    if (e == 0) { kmax[(long)b * g.C + c] = m; ksum[(long)b * g.C + c] = s; }
}

// ---- shared: crpe weight table for a channel chunk, zero-padded to 7x7, [cl][49] in LDS ---------
__device__ __forceinline__ void load_crpe_table(float* s_w, float* s_b, const FaGeom& g, const CrpeW& cw, int c0, int CC) {
    for (int i = threadIdx.x; i < CC * 49; i += blockDim.x) {
        const int cl = i / 49, t = i % 49, c = c0 + cl;
        float v = 0.f;
        if (c < g.C) {
            const int r = crpe_radius(g, c), di = t / 7 - 3, dj = t % 7 - 3;
            if (abs(di) <= r && abs(dj) <= r) v = crpe_wptr(g, cw, c, r)[(di + r) * (2 * r + 1) + (dj + r)];
        }
        s_w[i] = v;
    }
    for (int cl = threadIdx.x; cl < CC; cl += blockDim.x) {
        const int c = c0 + cl;
        s_b[cl] = c < g.C ? crpe_bias(g, cw, c, crpe_radius(g, c)) : 0.f;
    }
}

// ---- fwd C --------------------------------------------------------------------------------------
// block: CC channels x (256/CC) token lanes; grid.x over token groups, grid.y over channel chunks.
__global__ __launch_bounds__(256) void fa_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ Mmat,
                                                       const float* __restrict__ a, float* __restrict__ out, FaGeom g, CrpeW cw,
                                                       int CC, int tokens_per_block) {
    extern __shared__ float sm[];          // s_w[CC*49], s_b[CC]
    float* s_w = sm;
    float* s_b = sm + CC * 49;
    const int c0 = blockIdx.y * CC;
    load_crpe_table(s_w, s_b, g, cw, c0, CC);
    __syncthreads();
    const int cl = threadIdx.x % CC, tl = threadIdx.x / CC, ntl = blockDim.x / CC;
    const int c = c0 + cl;
    if (c >= g.C || tl >= ntl) return;
    const int head = c / g.Ch, ch = c % g.Ch, hb = head * g.Ch, r = crpe_radius(g, c);
    const int C3 = 3 * g.C;
    const long total = (long)g.B * g.N;
    const long t_beg = (long)blockIdx.x * tokens_per_block, t_end = min(total, t_beg + tokens_per_block);
    for (long tok = t_beg + tl; tok < t_end; tok += ntl) {
        const int b = (int)(tok / g.N), n = (int)(tok % g.N), h = n / g.W, w = n % g.W;
        const float* row = qkv + tok * C3;
        const float qc = row[c];
        float fa = 0.f;
        const float* Mb = Mmat + ((long)b * g.C + hb) * g.Ch + ch;
        for (int j = 0; j < g.Ch; ++j) fa = fmaf(row[hb + j], Mb[(long)j * g.Ch], fa);
        float u = s_b[cl];
        for (int di = -r; di <= r; ++di) {
            const int hh = h + di;
            if (hh < 0 || hh >= g.H) continue;
            for (int dj = -r; dj <= r; ++dj) {
                const int ww = w + dj;
                if (ww < 0 || ww >= g.W) continue;
                u = fmaf(s_w[cl * 49 + (di + 3) * 7 + (dj + 3)], qkv[((long)b * g.N + hh * g.W + ww) * C3 + 2 * g.C + c], u);
            }
        }
        float y = g.scale * fa + qc * u;
        if (a) y *= a[(long)b * g.C + c];
        out[tok * g.C + c] = y;
    }
}

// ---- bwd 1: reductions over tokens ---------------------------------------------------------------
// blockDim = TL * C (TL token lanes).  Thread = (token lane, channel c).  Per-thread register partials:
// da, d(bias), d(w[49]), dM[c][0..CH).  dFA rows and q rows of FA_TCHUNK tokens go through LDS so that a
// thread can see the other channels of its head.
template <int CH>
__global__ __launch_bounds__(512) void fa_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ qkv, const float* __restrict__ Mmat,
                                     const float* __restrict__ a, float* __restrict__ da, float* __restrict__ dM,
                                     float* dw3, float* db3, float* dw5, float* db5, float* dw7, float* db7,
                                     FaGeom g, CrpeW cw, int TL, int tokens_per_block) {
    extern __shared__ float sm[];           // s_dfa[TL][TCHUNK][C], s_q[TL][TCHUNK][C]
    const int C = g.C, C3 = 3 * C;
    float* s_dfa = sm;
    float* s_q = sm + TL * FA_TCHUNK * C;
    const int c = threadIdx.x % C, tl = threadIdx.x / C;
    const int b = blockIdx.y;
    const int head = c / CH, ch = c % CH, hb = head * CH, r = crpe_radius(g, c);
    const float* wp = crpe_wptr(g, cw, c, r);
    const float bias = crpe_bias(g, cw, c, r);
    const int win = 2 * r + 1;
    const float ac = a ? a[(long)b * C + c] : 1.f;
    const int n_beg = blockIdx.x * tokens_per_block, n_end = min(g.N, n_beg + tokens_per_block);
    float acc_da = 0.f, acc_db = 0.f;
    float acc_w[49], acc_m[CH];
#pragma unroll
    for (int t = 0; t < 49; ++t) acc_w[t] = 0.f;
#pragma unroll
    for (int e = 0; e < CH; ++e) acc_m[e] = 0.f;
    const float* Mb = Mmat + ((long)b * C + hb) * CH + ch;
    const int per_step = TL * FA_TCHUNK;
    for (int base = n_beg; base < n_end; base += per_step) {
#pragma unroll 1
        for (int tt = 0; tt < FA_TCHUNK; ++tt) {
            const int n = base + tl * FA_TCHUNK + tt;
            float dfa = 0.f, qc = 0.f;
            if (n < n_end) {
                const long tok = (long)b * g.N + n;
                const float* row = qkv + tok * C3;
                const float G = dout[tok * C + c];
                qc = row[c];
                const float dY = ac * G, dU = dY * qc;
                const int h = n / g.W, w = n % g.W;
                float u = bias;
#pragma unroll
                for (int i = 0; i < 7; ++i) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        if (i < win && j < win) {
                            const int hh = h + i - r, ww = w + j - r;
                            if (hh >= 0 && hh < g.H && ww >= 0 && ww < g.W) {
                                const float vn = qkv[((long)b * g.N + hh * g.W + ww) * C3 + 2 * C + c];
                                u = fmaf(wp[i * win + j], vn, u);
                                acc_w[i * 7 + j] = fmaf(dU, vn, acc_w[i * 7 + j]);
                            }
                        }
                    }
                }
                float fa = 0.f;
#pragma unroll
                for (int j = 0; j < CH; ++j) fa = fmaf(row[hb + j], Mb[j * CH], fa);
                const float Y = g.scale * fa + qc * u;
                acc_da = fmaf(G, Y, acc_da);
                acc_db += dU;
                dfa = g.scale * dY;
            }
            s_dfa[(tl * FA_TCHUNK + tt) * C + c] = dfa;
            s_q[(tl * FA_TCHUNK + tt) * C + c] = qc;
        }
        __syncthreads();
#pragma unroll 1
        for (int tt = 0; tt < FA_TCHUNK; ++tt) {
            const float qk = s_q[(tl * FA_TCHUNK + tt) * C + c];
            const float* drow = &s_dfa[(tl * FA_TCHUNK + tt) * C + hb];
#pragma unroll
            for (int e = 0; e < CH; ++e) acc_m[e] = fmaf(qk, drow[e], acc_m[e]);
        }
        __syncthreads();
    }
    if (a) atomicAdd(&da[(long)b * C + c], acc_da);
#pragma unroll
    for (int e = 0; e < CH; ++e) atomicAdd(&dM[((long)b * C + c) * CH + e], acc_m[e]);
    float* dwp; float* dbp;
    if (r == 1) { dwp = dw3 + (long)c * 9; dbp = db3 + c; }
    else if (r == 2) { dwp = dw5 + (long)(c - g.s3 * CH) * 25; dbp = db5 + (c - g.s3 * CH); }
    else { dwp = dw7 + (long)(c - (g.s3 + g.s5) * CH) * 49; dbp = db7 + (c - (g.s3 + g.s5) * CH); }
    atomicAdd(dbp, acc_db);
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (i < win && j < win) atomicAdd(&dwp[i * win + j], acc_w[i * 7 + j]);
}

// ---- bwd 2: t[b,c] = sum_e dM[b,c,e] * M[b,c,e] -----------------------------------------------
__global__ void fa_bwd_mid_kernel(const float* __restrict__ dM, const float* __restrict__ Mmat, float* __restrict__ tcol, int BC, int Ch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    float s = 0.f;
    for (int e = 0; e < Ch; ++e) s = fmaf(dM[(long)i * Ch + e], Mmat[(long)i * Ch + e], s);
    tcol[i] = s;
}

// ---- bwd 3: dq, dk, dv per (token, channel) ------------------------------------------------------
__global__ __launch_bounds__(256) void fa_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ qkv,
                                                           const float* __restrict__ Mmat, const float* __restrict__ a,
                                                           const float* __restrict__ kmax, const float* __restrict__ ksum,
                                                           const float* __restrict__ dM, const float* __restrict__ tcol,
                                                           float* __restrict__ dqkv, FaGeom g, CrpeW cw, int CC, int tokens_per_block) {
    extern __shared__ float sm[];
    float* s_w = sm;
    float* s_b = sm + CC * 49;
    const int c0 = blockIdx.y * CC;
    load_crpe_table(s_w, s_b, g, cw, c0, CC);
    __syncthreads();
    const int cl = threadIdx.x % CC, tl = threadIdx.x / CC, ntl = blockDim.x / CC;
    const int c = c0 + cl;
    if (c >= g.C || tl >= ntl) return;
    const int C = g.C, C3 = 3 * C, Ch = g.Ch;
    const int head = c / Ch, ch = c % Ch, hb = head * Ch, r = crpe_radius(g, c);
    const long total = (long)g.B * g.N;
    const long t_beg = (long)blockIdx.x * tokens_per_block, t_end = min(total, t_beg + tokens_per_block);
    for (long tok = t_beg + tl; tok < t_end; tok += ntl) {
        const int b = (int)(tok / g.N), n = (int)(tok % g.N), h = n / g.W, w = n % g.W;
        const float* row = qkv + tok * C3;
        const float* grow = dout + tok * C;
        const float* ab = a ? a + (long)b * C : nullptr;
        const float ac = ab ? ab[c] : 1.f;
        const float G = grow[c], qc = row[c], kc = row[C + c];
        const float dY = ac * G;
        // U (forward conv) and conv^T(dU) share the stencil walk
        float u = s_b[cl], dvc = 0.f;
        for (int di = -r; di <= r; ++di) {
            for (int dj = -r; dj <= r; ++dj) {
                const int hh = h + di, ww = w + dj;
                if (hh >= 0 && hh < g.H && ww >= 0 && ww < g.W)
                    u = fmaf(s_w[cl * 49 + (di + 3) * 7 + (dj + 3)], qkv[((long)b * g.N + hh * g.W + ww) * C3 + 2 * C + c], u);
                const int h2 = h - di, w2 = w - dj;      // token whose window position (di,dj) lands on n
                if (h2 >= 0 && h2 < g.H && w2 >= 0 && w2 < g.W) {
                    const long t2 = (long)b * g.N + h2 * g.W + w2;
                    dvc = fmaf(s_w[cl * 49 + (di + 3) * 7 + (dj + 3)], ac * dout[t2 * C + c] * qkv[t2 * C3 + c], dvc);
                }
            }
        }
        const float* Mrow = Mmat + ((long)b * C + c) * Ch;       // M[c][e]
        const float* dMrow = dM + ((long)b * C + c) * Ch;        // dM[c][e]
        float dq = 0.f, dP = 0.f, dv = dvc;
        for (int e = 0; e < Ch; ++e) {
            const float ae = ab ? ab[hb + e] : 1.f;
            dq = fmaf(g.scale * ae * grow[hb + e], Mrow[e], dq);                 // dFA[n,hb+e] * M[c][e]
            dP = fmaf(row[2 * C + hb + e], dMrow[e], dP);                         // v[n,hb+e] * dM[c][e]
            const float pj = expf(row[C + hb + e] - kmax[(long)b * C + hb + e]) / ksum[(long)b * C + hb + e];
            dv = fmaf(pj, dM[((long)b * C + hb + e) * Ch + ch], dv);              // P[n,hb+e] * dM[hb+e][ch]
        }
        dq = fmaf(dY, u, dq);
        const float P = expf(kc - kmax[(long)b * C + c]) / ksum[(long)b * C + c];
        float* drow = dqkv + tok * C3;
        drow[c] = dq;
        drow[C + c] = P * (dP - tcol[(long)b * C + c]);
        drow[2 * C + c] = dv;
    }
}

// ---- Domain Adapter -----------------------------------------------------------------------------
// grid = B; h1 = relu(W1 label + b1) in LDS, z = W2 h1 + b2 in LDS, a = softmax over heads per ch.
__global__ __launch_bounds__(256) void da_fwd_kernel(const float* __restrict__ label, const float* __restrict__ W1, const float* __restrict__ b1,
                                                     const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ a,
                                                     int D, int hid, int C, int heads) {
    extern __shared__ float sm[];   // h1[hid], z[C]
    float* h1 = sm;
    float* z = sm + hid;
    const int b = blockIdx.x, Ch = C / heads;
    for (int i = threadIdx.x; i < hid; i += blockDim.x) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(label[(long)b * D + d], W1[(long)i * D + d], s);
        h1[i] = fmaxf(s + b1[i], 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < hid; ++i) s = fmaf(h1[i], W2[(long)c * hid + i], s);
        z[c] = s + b2[c];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int ch = c % Ch;
        float m = -INFINITY;
        for (int hh = 0; hh < heads; ++hh) m = fmaxf(m, z[hh * Ch + ch]);
        float s = 0.f;
        for (int hh = 0; hh < heads; ++hh) s += expf(z[hh * Ch + ch] - m);
        a[(long)b * C + c] = expf(z[c] - m) / s;
    }
}

__global__ __launch_bounds__(256) void da_bwd_kernel(const float* __restrict__ label, const float* __restrict__ W1, const float* __restrict__ b1,
                                                     const float* __restrict__ W2, const float* __restrict__ a, const float* __restrict__ da,
                                                     float* dW1, float* db1, float* dW2, float* db2, int D, int hid, int C, int heads) {
    extern __shared__ float sm[];   // h1[hid], dz[C], dh[hid]
    float* h1 = sm;
    float* dz = sm + hid;
    float* dh = dz + C;
    const int b = blockIdx.x, Ch = C / heads;
    for (int i = threadIdx.x; i < hid; i += blockDim.x) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s = fmaf(label[(long)b * D + d], W1[(long)i * D + d], s);
        h1[i] = s + b1[i];          // pre-ReLU
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int ch = c % Ch;
        float dot = 0.f;
        for (int hh = 0; hh < heads; ++hh) dot = fmaf(a[(long)b * C + hh * Ch + ch], da[(long)b * C + hh * Ch + ch], dot);
        dz[c] = a[(long)b * C + c] * (da[(long)b * C + c] - dot);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        atomicAdd(&db2[c], dz[c]);
        for (int i = 0; i < hid; ++i) atomicAdd(&dW2[(long)c * hid + i], dz[c] * fmaxf(h1[i], 0.f));
    }
    for (int i = threadIdx.x; i < hid; i += blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(dz[c], W2[(long)c * hid + i], s);
        dh[i] = h1[i] > 0.f ? s : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hid; i += blockDim.x) {
        atomicAdd(&db1[i], dh[i]);
        for (int d = 0; d < D; ++d) atomicAdd(&dW1[(long)i * D + d], dh[i] * label[(long)b * D + d]);
    }
}

bool make_geom(FaGeom& g, int B, int H, int W, int C, int heads, int s3, int s5, int s7) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0 || C % heads != 0) return false;
    if (s3 + s5 + s7 != heads) return false;
    g.B = B; g.H = H; g.W = W; g.N = H * W; g.C = C; g.heads = heads; g.Ch = C / heads; g.s3 = s3; g.s5 = s5; g.s7 = s7;
    g.scale = 1.0f / sqrtf((float)g.Ch);
    return true;
}

int fa_cw(const FaGeom& g) {
    int hp = 64 / g.Ch;
    if (hp < 1) hp = 1;
    if (hp > g.heads) hp = g.heads;
    while (g.heads % hp) --hp;
    return hp * g.Ch;
}

size_t fa_ws_floats(int B, int N, int C, int heads) {
    const int Ch = C / heads;
    const long NT = (N + FA_T - 1) / FA_T;
    const long fwd = (long)B * NT * C * (2 + Ch);
    const long bwd = (long)B * C * (1 + Ch);
    return (size_t)(fwd > bwd ? fwd : bwd);
}

}  // namespace

extern "C" size_t mdvit_factoratt_ws_bytes(int32_t B, int32_t N, int32_t C, int32_t heads) {
    if (B <= 0 || N <= 0 || C <= 0 || heads <= 0 || C % heads) return 0;
    return fa_ws_floats(B, N, C, heads) * sizeof(float);
}

extern "C" int mdvit_factoratt_fwd(const float* qkv, const float* w3, const float* b3, const float* w5, const float* b5,
                                   const float* w7, const float* b7, const float* a, float* out, float* kmax, float* ksum, float* Mmat,
                                   void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                                   int32_t s3, int32_t s5, int32_t s7, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    FaGeom g;
    MDVIT_CHECK_ARG(make_geom(g, B, H, W, C, heads, s3, s5, s7), MDVIT_E_SHAPE, "factoratt_fwd: bad geometry B=%d H=%d W=%d C=%d heads=%d splits=%d/%d/%d", B, H, W, C, heads, s3, s5, s7);
    MDVIT_CHECK_ARG(ws_bytes >= fa_ws_floats(B, g.N, C, heads) * sizeof(float), MDVIT_E_WORKSPACE, "factoratt_fwd: workspace too small (%zu bytes)", ws_bytes);
    const int NT = cdiv(g.N, FA_T), CW = fa_cw(g);
    MDVIT_CHECK_ARG(CW <= 128, MDVIT_E_SHAPE, "factoratt_fwd: head dim %d too large", g.Ch);
    float* ws_m = (float*)ws;
    float* ws_s = ws_m + (long)B * NT * C;
    float* ws_P = ws_s + (long)B * NT * C;
    hipLaunchKernelGGL(fa_kv_partial_kernel, dim3(NT, C / CW, B), dim3(256), sizeof(float) * 2 * FA_T * CW, s, qkv, ws_m, ws_s, ws_P, g, CW, NT);
    hipLaunchKernelGGL(fa_kv_combine_kernel, dim3(cdiv((long)C * g.Ch, 256), B), dim3(256), 0, s, ws_m, ws_s, ws_P, kmax, ksum, Mmat, g, NT);
    CrpeW cw{w3, b3, w5, b5, w7, b7};
    const int CC = C < 128 ? C : 128;          // channel chunk; 256 % CC == 0 for C in {64,128,...}
    int block = 256;
    if (256 % CC) block = CC * (256 / CC > 0 ? 256 / CC : 1);
    const long total = (long)B * g.N;
    int tpb = (int)max(32L, (total + 2047) / 2048);
    hipLaunchKernelGGL(fa_apply_kernel, dim3(cdiv(total, tpb), cdiv(C, CC)), dim3(block), sizeof(float) * CC * 50, s, qkv, Mmat, a, out, g, cw, CC, tpb);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_factoratt_bwd(const float* dout, const float* qkv, const float* w3, const float* b3, const float* w5, const float* b5,
                                   const float* w7, const float* b7, const float* a, const float* kmax, const float* ksum, const float* Mmat,
                                   float* dqkv, float* da, float* dw3, float* db3, float* dw5, float* db5, float* dw7, float* db7,
                                   void* ws, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads,
                                   int32_t s3, int32_t s5, int32_t s7, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    FaGeom g;
    MDVIT_CHECK_ARG(make_geom(g, B, H, W, C, heads, s3, s5, s7), MDVIT_E_SHAPE, "factoratt_bwd: bad geometry B=%d H=%d W=%d C=%d heads=%d", B, H, W, C, heads);
    MDVIT_CHECK_ARG(ws_bytes >= fa_ws_floats(B, g.N, C, heads) * sizeof(float), MDVIT_E_WORKSPACE, "factoratt_bwd: workspace too small (%zu bytes)", ws_bytes);
    MDVIT_CHECK_ARG(C <= 512, MDVIT_E_SHAPE, "factoratt_bwd: C=%d > 512 not built", C);
    MDVIT_CHECK_ARG((a == nullptr) == (da == nullptr), MDVIT_E_SHAPE, "factoratt_bwd: a and da must both be given or both be NULL");
    float* tcol = (float*)ws;
    float* dM = tcol + (long)B * C;
    const int Ch = g.Ch;
    MDVIT_ZERO(dM, sizeof(float) * (size_t)B * C * Ch, s);
    if (da) MDVIT_ZERO(da, sizeof(float) * (size_t)B * C, s);
    MDVIT_ZERO(dw3, sizeof(float) * s3 * Ch * 9, s);  MDVIT_ZERO(db3, sizeof(float) * s3 * Ch, s);
    MDVIT_ZERO(dw5, sizeof(float) * s5 * Ch * 25, s); MDVIT_ZERO(db5, sizeof(float) * s5 * Ch, s);
    MDVIT_ZERO(dw7, sizeof(float) * s7 * Ch * 49, s); MDVIT_ZERO(db7, sizeof(float) * s7 * Ch, s);
    CrpeW cw{w3, b3, w5, b5, w7, b7};
    const int TL = max(1, 256 / C);
    const int block = TL * C;
    int tpb = TL * FA_TCHUNK * max(1, 128 / (TL * FA_TCHUNK));       // ~128 tokens per block
    const size_t lds = sizeof(float) * 2 * TL * FA_TCHUNK * C;
    dim3 grid(cdiv(g.N, tpb), B);
#define FA_BWD_LAUNCH(CHV) hipLaunchKernelGGL((fa_bwd_reduce_kernel<CHV>), grid, dim3(block), lds, s, dout, qkv, Mmat, a, da, dM, dw3, db3, dw5, db5, dw7, db7, g, cw, TL, tpb)
    switch (Ch) {
        case 8: FA_BWD_LAUNCH(8); break;
        case 16: FA_BWD_LAUNCH(16); break;
        case 40: FA_BWD_LAUNCH(40); break;
        case 64: FA_BWD_LAUNCH(64); break;
        default: return mdvit_set_error(MDVIT_E_SHAPE, "factoratt_bwd: head dim %d not built (8/16/40/64)", Ch);
    }
#undef FA_BWD_LAUNCH
    hipLaunchKernelGGL(fa_bwd_mid_kernel, dim3(cdiv((long)B * C, 256)), dim3(256), 0, s, dM, Mmat, tcol, B * C, Ch);
    const int CC = C < 128 ? C : 128;
    int blk = 256;
    if (256 % CC) blk = CC * (256 / CC > 0 ? 256 / CC : 1);
    const long total = (long)B * g.N;
    int tpb2 = (int)max(32L, (total + 2047) / 2048);
    hipLaunchKernelGGL(fa_bwd_apply_kernel, dim3(cdiv(total, tpb2), cdiv(C, CC)), dim3(blk), sizeof(float) * CC * 50, s,
                       dout, qkv, Mmat, a, kmax, ksum, dM, tcol, dqkv, g, cw, CC, tpb2);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_da_fwd(const float* label, const float* W1, const float* b1, const float* W2, const float* b2, float* a,
                            int32_t B, int32_t D, int32_t hid, int32_t C, int32_t heads, void* stream) {
    MDVIT_CHECK_ARG(B > 0 && D > 0 && hid > 0 && C > 0 && heads > 0 && C % heads == 0, MDVIT_E_SHAPE, "da_fwd: bad shape");
    hipLaunchKernelGGL(da_fwd_kernel, dim3(B), dim3(256), sizeof(float) * (hid + C), (hipStream_t)stream, label, W1, b1, W2, b2, a, D, hid, C, heads);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}

extern "C" int mdvit_da_bwd(const float* label, const float* W1, const float* b1, const float* W2, const float* b2, const float* a,
                            const float* da, float* dW1, float* db1, float* dW2, float* db2,
                            int32_t B, int32_t D, int32_t hid, int32_t C, int32_t heads, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    (void)b2;
    MDVIT_CHECK_ARG(B > 0 && D > 0 && hid > 0 && C > 0 && heads > 0 && C % heads == 0, MDVIT_E_SHAPE, "da_bwd: bad shape");
    MDVIT_ZERO(dW1, sizeof(float) * hid * D, s); MDVIT_ZERO(db1, sizeof(float) * hid, s);
    MDVIT_ZERO(dW2, sizeof(float) * (size_t)C * hid, s); MDVIT_ZERO(db2, sizeof(float) * C, s);
    hipLaunchKernelGGL(da_bwd_kernel, dim3(B), dim3(256), sizeof(float) * (2 * hid + C), s, label, W1, b1, W2, a, da, dW1, db1, dW2, db2, D, hid, C, heads);
    MDVIT_LAUNCH_CHECK();
    return MDVIT_OK;
}
