// HBM bandwidth probe: what do streaming float4 reads / writes / copies reach on this chip, with plain and
// non-temporal accesses?  (sets the practical roof for the store-bound GEMM epilogues)   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT> __global__ __launch_bounds__(256) void fill_k(f4* __restrict__ p, size_t n) {
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
template <int NT> __global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ a, f4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
template <int NT> __global__ __launch_bounds__(256) void read_k(const f4* __restrict__ a, float* out, size_t n) {
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        s += v;
    }
    if (s.x + s.y + s.z + s.w == 123.456f) out[0] = 1.f;
}
// tile-shaped stores like a GEMM epilogue: each workgroup writes a [128 rows][128 cols] tile of an [M][N] matrix
template <int NT> __global__ __launch_bounds__(256) void tile_k(float* __restrict__ p, int M, int N) {
    const int tn = N / 128;
    const int tm = blockIdx.x / tn, tc = blockIdx.x % tn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (int r = 0; r < 16; ++r) {
        const int row = tm * 128 + wave * 32 + r * 2 + (lane >> 5);
        f4* dst = (f4*)(p + (size_t)row * N + tc * 128 + (lane & 31) * 4);
        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
}

// the GEMM epilogue's current pattern: per store instruction 32 rows x 32 bytes (lane%32 = row, lane/32 = 16-byte half),
// four instructions cover a 32x32 block; a wave owns WTM x WTN such blocks of a 128x128 (2x2 waves) tile
template <int NT> __global__ __launch_bounds__(256) void mfma_pattern_k(float* __restrict__ p, int M, int N) {
    const int tn = N / 128;
    const int tm = blockIdx.x / tn, tc = blockIdx.x % tn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int q = 0; q < 4; ++q) {
                const int row = tm * 128 + wm0 + i * 32 + (lane & 31);
                const int col = tc * 128 + wn0 + j * 32 + 8 * q + 4 * (lane >> 5);
                f4* dst = (f4*)(p + (size_t)row * N + col);
                if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
}
// the same tile written row-contiguously: per instruction 8 lanes x 16 B = 128 B of one row, 8 rows (what an LDS-transposed
// epilogue of a 32-column wave block can do), or 16 lanes = 256 B x 4 rows for a 64-column wave block
template <int LPR> __global__ __launch_bounds__(256) void rowseg_k(float* __restrict__ p, int M, int N) {
    const int tn = N / 128;
    const int tm = blockIdx.x / tn, tc = blockIdx.x % tn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    constexpr int RPI = 64 / LPR;            // rows per instruction
    constexpr int WCOLS = LPR * 4;           // columns per instruction
    for (int cb = 0; cb < 64 / WCOLS; ++cb)
        for (int r = 0; r < 64 / RPI; ++r) {
            const int row = tm * 128 + wm0 + r * RPI + lane / LPR;
            const int col = tc * 128 + wn0 + cb * WCOLS + (lane % LPR) * 4;
            *(f4*)(p + (size_t)row * N + col) = v;
        }
}

// GEMM-shaped memory phases without the arithmetic: a 128x128 output tile of C[M][N] from A[M][64] (K = 64):
// load the A rows (32 KB) and 32 KB of "weights" (L2-resident), barrier, then store the tile in the MFMA pattern.
// PERSIST: a fixed grid walks the tiles and issues the next tile's loads before storing the current one.
template <int PERSIST> __global__ __launch_bounds__(256) void gemm_phases_k(const float* __restrict__ A, const float* __restrict__ W,
                                                                            float* __restrict__ p, int M, int N, int ntiles) {
    __shared__ f4 sm[2048];
    const int tn = N / 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    f4 ra[8], rw[8];
    auto load = [&](int tile) {
        const int tm = tile / tn, tc = tile % tn;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int idx = v * 256 + threadIdx.x;         // float4 index inside the [128][64] A block
            ra[v] = *(const f4*)(A + (size_t)(tm * 128 + idx / 16) * 64 + (idx % 16) * 4);
            rw[v] = *(const f4*)(W + (size_t)(tc * 128 + idx / 16) * 64 + (idx % 16) * 4);
        }
    };
    auto store = [&](int tile, f4 v) {
        const int tm = tile / tn, tc = tile % tn;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = tm * 128 + wm0 + i * 32 + (lane & 31);
                    const int col = tc * 128 + wn0 + j * 32 + 8 * q + 4 * (lane >> 5);
                    *(f4*)(p + (size_t)row * N + col) = v;
                }
    };
    if (!PERSIST) {
        load(blockIdx.x);
        f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 8; ++v) sm[v * 256 + threadIdx.x] = ra[v] + rw[v];
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 8; ++v) acc += sm[(v * 256 + threadIdx.x * 7) & 2047];
        store(blockIdx.x, acc);
    } else {
        int tile = blockIdx.x;
        if (tile < ntiles) load(tile);
        for (; tile < ntiles; tile += gridDim.x) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < 8; ++v) sm[v * 256 + threadIdx.x] = ra[v] + rw[v];
            __syncthreads();
            if (tile + (int)gridDim.x < ntiles) load(tile + gridDim.x);
#pragma unroll
            for (int v = 0; v < 8; ++v) acc += sm[(v * 256 + threadIdx.x * 7) & 2047];
            store(tile, acc);
            __syncthreads();
        }
    }
}

int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    f4 *a, *b; float* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, double gb, auto&& fn) {
        fn(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) fn();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f us  %6.2f TB/s\n", name, ms * 100.0, gb / (ms / 10.0 * 1e-3) / 1e3);
    };
    for (int grid : {2048, 8192, 65536}) {
        printf("grid %d\n", grid);
        time("fill plain", 1.0737, [&] { hipLaunchKernelGGL(fill_k<0>, dim3(grid), dim3(256), 0, 0, a, n); });
        time("fill nt", 1.0737, [&] { hipLaunchKernelGGL(fill_k<1>, dim3(grid), dim3(256), 0, 0, a, n); });
        time("copy plain", 2.1475, [&] { hipLaunchKernelGGL(copy_k<0>, dim3(grid), dim3(256), 0, 0, a, b, n); });
        time("copy nt", 2.1475, [&] { hipLaunchKernelGGL(copy_k<1>, dim3(grid), dim3(256), 0, 0, a, b, n); });
        time("read plain", 1.0737, [&] { hipLaunchKernelGGL(read_k<0>, dim3(grid), dim3(256), 0, 0, a, o, n); });
        time("read nt", 1.0737, [&] { hipLaunchKernelGGL(read_k<1>, dim3(grid), dim3(256), 0, 0, a, o, n); });
    }
    const int M = 524288, N = 512;
    time("tile stores plain [M,512]", 1.0737, [&] { hipLaunchKernelGGL(tile_k<0>, dim3((M / 128) * (N / 128)), dim3(256), 0, 0, (float*)a, M, N); });
    time("tile stores nt    [M,512]", 1.0737, [&] { hipLaunchKernelGGL(tile_k<1>, dim3((M / 128) * (N / 128)), dim3(256), 0, 0, (float*)a, M, N); });
    {
        const int Mm = 262144, Nn = 512, nt = (Mm / 128) * (Nn / 128);
        const double gb = 4.0 * ((double)Mm * 64 + (double)Mm * Nn) / 1e9;
        time("gemm phases, 1 tile per WG", gb, [&] { hipLaunchKernelGGL(gemm_phases_k<0>, dim3(nt), dim3(256), 0, 0, (const float*)b, (const float*)b, (float*)a, Mm, Nn, nt); });
        for (int g : {256, 512, 1024, 2048})  {
            char nm[64]; snprintf(nm, 64, "gemm phases persistent g=%d", g);
            time(nm, gb, [&] { hipLaunchKernelGGL(gemm_phases_k<1>, dim3(g), dim3(256), 0, 0, (const float*)b, (const float*)b, (float*)a, Mm, Nn, nt); });
        }
    }
    for (int Nn : {512}) {
        const int Mm = (int)(((size_t)1 << 28) / Nn);
        char nm[64];
        snprintf(nm, 64, "mfma pattern   N=%d", Nn);
        time(nm, 1.0737, [&] { hipLaunchKernelGGL(mfma_pattern_k<0>, dim3((Mm / 128) * (Nn / 128)), dim3(256), 0, 0, (float*)a, Mm, Nn); });
        snprintf(nm, 64, "mfma pattern nt N=%d", Nn);
        time(nm, 1.0737, [&] { hipLaunchKernelGGL(mfma_pattern_k<1>, dim3((Mm / 128) * (Nn / 128)), dim3(256), 0, 0, (float*)a, Mm, Nn); });
        snprintf(nm, 64, "rows 128B x8   N=%d", Nn);
        time(nm, 1.0737, [&] { hipLaunchKernelGGL(rowseg_k<8>, dim3((Mm / 128) * (Nn / 128)), dim3(256), 0, 0, (float*)a, Mm, Nn); });
        snprintf(nm, 64, "rows 256B x4   N=%d", Nn);
        time(nm, 1.0737, [&] { hipLaunchKernelGGL(rowseg_k<16>, dim3((Mm / 128) * (Nn / 128)), dim3(256), 0, 0, (float*)a, Mm, Nn); });
    }
    return 0;
}
