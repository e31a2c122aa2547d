// what v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) do, lane by lane: hipcc --offload-arch=gfx950 permlane_probe.hip -o /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x * 4 + 0] = r[0]; out[threadIdx.x * 4 + 1] = r[1];
    out[threadIdx.x * 4 + 2] = s[0]; out[threadIdx.x * 4 + 3] = s[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 8) printf("lane %2d: swap16 -> (%4u, %4u)   swap32 -> (%4u, %4u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
