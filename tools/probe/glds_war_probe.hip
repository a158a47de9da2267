// Probe (gfx950): is the address VGPR pair of a global_load_lds_dwordx4 safe to overwrite in the very next instruction?
// Every wave issues NP back-to-back LDS-DMA pieces from a cold source; right after each one a VALU instruction overwrites the address pair with the
// address of a POISON buffer.  If the instruction still reads its address registers after issue, poison shows up in LDS.
// mode 0: v_mov_b64 over the pair immediately; mode 1: the same after s_nop 7; mode 2: no overwrite (control); mode 3: v_add_u32 on the low half only.
// Build: hipcc --offload-arch=gfx950 -O2 -w tools/probe/glds_war_probe.hip -o tools/probe/glds_war_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

constexpr int NP = 64;
template <int MODE>
__global__ __launch_bounds__(256) void war_test(const uint32_t* __restrict__ src, const uint32_t* __restrict__ poison, int* __restrict__ bad, int* __restrict__ badlane) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NP * 256; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const uint32_t* s0 = src + (long)blockIdx.x * NP * 256;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
    for (int k = 0; k < NP / 4; ++k) {
        const int pc = wave + 4 * k;
        uint64_t a = (uint64_t)(s0 + pc * 256 + lane * 4);
        const uint64_t j = (uint64_t)(poison + lane * 4);
        const uint32_t m = __builtin_amdgcn_readfirstlane(lds0 + pc * 1024);
        if (MODE == 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tv_mov_b64 %0, %1\n" : "+v"(a) : "v"(j), "s"(m) : "memory", "m0");
        if (MODE == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_nop 7\n\tv_mov_b64 %0, %1\n" : "+v"(a) : "v"(j), "s"(m) : "memory", "m0");
        if (MODE == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n" : "+v"(a) : "v"(j), "s"(m) : "memory", "m0");
        if (MODE == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tv_lshl_add_u64 %0, %1, 0, 0\n" : "+v"(a) : "v"(j), "s"(m) : "memory", "m0");
        if (a == 1) bad[0] = 1;      // keep `a` alive
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < NP * 256; i += 256) {
        const uint32_t v = reinterpret_cast<uint32_t*>(smem)[i];
        if (v != s0[i]) { atomicAdd(&bad[1 + (v == 0xbad0bad0u ? 0 : 1)], 1); atomicAdd(&badlane[(i % 256) / 4], 1); }
    }
}

template <int MODE>
static void run(const uint32_t* src, const uint32_t* poison, int* bad, int* badlane, int nwg) {
    hipMemset(bad, 0, 64); hipMemset(badlane, 0, 256);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&war_test<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, NP * 1024);
    hipLaunchKernelGGL((war_test<MODE>), dim3(nwg), dim3(256), NP * 1024, 0, src, poison, bad, badlane);
    hipDeviceSynchronize();
    int hb[16], hl[64];
    hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost); hipMemcpy(hl, badlane, 256, hipMemcpyDeviceToHost);
    printf("mode %d: wrong dwords: %d poison, %d other; per lane:", MODE, hb[1], hb[2]);
    for (int l = 0; l < 64; ++l) if (hl[l]) printf(" l%d:%d", l, hl[l]);
    printf("\n");
}

int main() {
    const int nwg = 2048;
    uint32_t *src, *poison; int *bad, *badlane;
    hipMalloc(&src, (size_t)nwg * NP * 1024 + (2 << 20)); hipMalloc(&poison, 4096); hipMalloc(&bad, 64); hipMalloc(&badlane, 256);
    std::vector<uint32_t> hs((size_t)nwg * NP * 256 + (512 << 10)), hp(1024, 0xbad0bad0u);
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = 0x10000u + (uint32_t)i;
    hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(poison, hp.data(), 4096, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<2>(src, poison, bad, badlane, nwg);
        run<0>(src, poison, bad, badlane, nwg);
        run<1>(src, poison, bad, badlane, nwg);
        run<3>(src, poison, bad, badlane, nwg);
    }
    return 0;
}
