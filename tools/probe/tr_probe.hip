// Probe (gfx950): (1) lane/element mapping of ds_read_b64_tr_b16, (2) LDS destination rule of global_load_lds_dwordx4.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/tr_probe.hip -o tools/probe/tr_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void tr_kernel(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane i -> element 4*i (linear 8-byte pieces).  mode 1: [t][n] image with row stride 64 elements:
    // 16-lane group g reads a 4(t) x 16(n) block: lane i -> &L[tb + (i>>2)][nb + (i&3)*4]
    int elem;
    if (mode == 0) elem = l * 4;
    else {
        const int g = l >> 4, i = l & 15;
        const int tb = 8 * (g >> 1), nb = 16 * (g & 1);
        elem = (tb + (i >> 2)) * 64 + nb + (i & 3) * 4;
    }
    const v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(lds + elem));
    out[l * 4 + 0] = v[0]; out[l * 4 + 1] = v[1]; out[l * 4 + 2] = v[2]; out[l * 4 + 3] = v[3];
}

__global__ void glds_kernel(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const int l = threadIdx.x;
    // each lane fetches 16 B from a PERMUTED global location; LDS destination should be base + lane*16
    const uint32_t* g = src + 4 * ((l * 7) & 63);
    __builtin_amdgcn_global_load_lds(g, lds + 64, 16, 0, 0);       // base = &lds[64]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    std::vector<uint16_t> h(256);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("tr mode %d:\n", mode);
        for (int l = 0; l < 64; ++l) printf(" lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    }
    uint32_t *s, *o; hipMalloc(&s, 4096); hipMalloc(&o, 4096);
    std::vector<uint32_t> hs(1024), ho(1024);
    for (int i = 0; i < 1024; ++i) hs[i] = i;
    hipMemcpy(s, hs.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(glds_kernel, dim3(1), dim3(64), 0, 0, s, o);
    hipMemcpy(ho.data(), o, 4096, hipMemcpyDeviceToHost);
    printf("glds: lds[60..72] = ");
    for (int i = 60; i < 72; ++i) printf("%x ", ho[i]);
    printf("\n");
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (ho[64 + 4 * l + j] != (uint32_t)(4 * ((l * 7) & 63) + j)) ok = 0;
    printf("glds dest = base + lane*16 with per-lane source: %s\n", ok ? "CONFIRMED" : "NO");
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return 0;
}
