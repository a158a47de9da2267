// Probe (gfx950): does a 1 KiB global_load_lds wave-instruction land correctly whatever the workgroup's LDS base is?
// A "blocker" kernel (small static LDS, spins) occupies the bottom of every CU's LDS on stream B; the test kernel (72 KiB dynamic LDS, every 1 KiB piece
// filled by one global_load_lds_dwordx4 and read back with ds_read) runs next to it on stream A, so its workgroups get LDS bases that are NOT multiples
// of 1 KiB and some piece straddles an absolute 64 KiB boundary of the CU's LDS.  Prints the LDS_ALLOC register of the workgroups and every piece that
// read back wrong.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/glds_base_probe.hip -o tools/probe/glds_base_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <map>

template <int BYTES>
__global__ void blocker(long long cycles, int* sink) {
    __shared__ int s[BYTES / 4];
    s[threadIdx.x % (BYTES / 4)] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    int v = 0;
    while (wall_clock64() - t0 < cycles) v += s[(threadIdx.x + v) % (BYTES / 4)];
    if (v == 0x7fffffff) sink[0] = v;
}

constexpr int NP = 72;       // pieces of 1 KiB
__global__ __launch_bounds__(256) void glds_test(const uint32_t* __restrict__ src, int* __restrict__ bad, uint32_t* __restrict__ regs, int rounds) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) regs[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 6);      // HW_REG_LDS_ALLOC
    for (int r = 0; r < rounds; ++r) {
        for (int i = tid; i < NP * 256; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0xdeadbeefu;
        __syncthreads();
        for (int pc = wave; pc < NP; pc += 4)
            __builtin_amdgcn_global_load_lds(src + (long)pc * 256 + lane * 4, (__attribute__((address_space(3))) void*)(smem + pc * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = tid; i < NP * 256; i += 256) {
            const uint32_t v = reinterpret_cast<uint32_t*>(smem)[i];
            if (v != src[i]) atomicAdd(&bad[blockIdx.x * NP + i / 256], 1);
        }
        __syncthreads();
    }
}

template <int BYTES>
static void run(const uint32_t* src, int* bad, uint32_t* regs, int* sink, hipStream_t sA, hipStream_t sB, int nwg, bool with_blocker) {
    hipMemset(bad, 0, sizeof(int) * nwg * NP);
    hipDeviceSynchronize();
    if (with_blocker) hipLaunchKernelGGL((blocker<BYTES>), dim3(256), dim3(64), 0, sB, 400000LL, sink);     // ~4 ms at 100 MHz wall clock
    hipLaunchKernelGGL(glds_test, dim3(nwg), dim3(256), NP * 1024, sA, src, bad, regs, 4);
    hipDeviceSynchronize();
    std::vector<int> hb(nwg * NP);
    std::vector<uint32_t> hr(nwg);
    hipMemcpy(hb.data(), bad, sizeof(int) * nwg * NP, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), regs, sizeof(uint32_t) * nwg, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> hist;
    for (int w = 0; w < nwg; ++w) hist[hr[w]]++;
    printf("blocker LDS %d B (%s): LDS_ALLOC register values:", BYTES, with_blocker ? "running" : "absent");
    for (auto& kv : hist) printf(" 0x%08x x%d", kv.first, kv.second);
    printf("\n");
    int nbad = 0;
    for (int w = 0; w < nwg; ++w)
        for (int p = 0; p < NP; ++p)
            if (hb[w * NP + p]) { if (nbad < 24) printf("   WG %d (LDS_ALLOC 0x%08x): piece %d, %d wrong dwords (of 4 x 256)\n", w, hr[w], p, hb[w * NP + p]); ++nbad; }
    printf("   wrong pieces: %d\n", nbad);
}

int main() {
    uint32_t* src; int* bad; uint32_t* regs; int* sink;
    const int nwg = 1024;
    hipMalloc(&src, NP * 1024); hipMalloc(&bad, sizeof(int) * nwg * NP); hipMalloc(&regs, sizeof(uint32_t) * nwg); hipMalloc(&sink, 64);
    std::vector<uint32_t> hs(NP * 256);
    for (int i = 0; i < NP * 256; ++i) hs[i] = 0x10000u + i;
    hipMemcpy(src, hs.data(), NP * 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&glds_test), hipFuncAttributeMaxDynamicSharedMemorySize, NP * 1024);
    hipStream_t sA, sB; hipStreamCreate(&sA); hipStreamCreate(&sB);
    run<512>(src, bad, regs, sink, sA, sB, nwg, false);
    run<512>(src, bad, regs, sink, sA, sB, nwg, true);
    run<1280>(src, bad, regs, sink, sA, sB, nwg, true);
    run<1536>(src, bad, regs, sink, sA, sB, nwg, true);
    run<2304>(src, bad, regs, sink, sA, sB, nwg, true);
    run<4608>(src, bad, regs, sink, sA, sB, nwg, true);
    run<1024>(src, bad, regs, sink, sA, sB, nwg, true);
    return 0;
}
