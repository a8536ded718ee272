// (GPU probe, not product code) what does global_load_lds_dwordx3 do on gfx950: per-lane LDS stride, behaviour under an exec mask, and whether it observes the
// same wave's earlier stores to the same addresses.     hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_probe scripts/probes/dma_probe.hip && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;
__global__ __launch_bounds__(64) void probe(uint32_t* g, uint32_t* out, int mode) {
    __shared__ uint32_t buf[64 * 4 + 64];
    const uint32_t lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 + 64; i += 64) buf[i] = 0xdeadbeefu;
    __syncthreads();
    if (mode == 1) {                      // store first, then DMA the same addresses without a wait in between
        g[3 * lane] = 1000 + lane; g[3 * lane + 1] = 2000 + lane; g[3 * lane + 2] = 3000 + lane;
    }
    if (mode == 2) { if (lane & 1) __builtin_amdgcn_global_load_lds((gbl_cvoid*)(g + 3 * lane), (lds_void*)buf, 12, 0, 0); }
    else __builtin_amdgcn_global_load_lds((gbl_cvoid*)(g + 3 * lane), (lds_void*)buf, 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 64 * 4 + 64; i += 64) out[i] = buf[i];
}
int main() {
    uint32_t *g, *out;
    hipMalloc(&g, 4096 * 4); hipMalloc(&out, 4096 * 4);
    std::vector<uint32_t> h(4096), o(4096);
    for (int mode = 0; mode < 3; ++mode) {
        for (int i = 0; i < 4096; ++i) h[i] = i;
        hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, out, mode);
        hipMemcpy(o.data(), out, 320 * 4, hipMemcpyDeviceToHost);
        printf("mode %d:", mode);
        for (int i = 0; i < 40; ++i) printf(" %u", o[i]);
        printf(" ... [189..200]:");
        for (int i = 189; i < 200; ++i) printf(" %u", o[i]);
        printf("\n");
    }
    return 0;
}
