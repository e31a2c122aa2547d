// do v_cvt_f16_f32 and v_cvt_pk_f16_f32 agree on gfx950 (denormal halves, ties)?  build: hipcc --offload-arch=gfx950 tools/cvt_probe.hip -o /tmp/cvt_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, uint16_t* a, uint16_t* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i];
    _Float16 s;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(s) : "v"(v));
    a[i] = __builtin_bit_cast(uint16_t, s);
    uint32_t p;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(v), "v"(v));
    b[i] = (uint16_t)(p & 0xffffu);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n);
    uint32_t st = 12345;
    for (int i = 0; i < n; i++) {
        st = st * 1664525u + 1013904223u;
        uint32_t bits;
        if (i & 1) bits = (st & 0x807fffffu) | ((uint32_t)(96 + (st >> 24) % 40) << 23);   // small magnitudes around the half denormal range
        else bits = st;
        std::memcpy(&x[i], &bits, 4);
    }
    float* dx; uint16_t *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 2); hipMalloc(&db, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, n);
    std::vector<uint16_t> a(n), b(n);
    hipMemcpy(a.data(), da, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 2, hipMemcpyDeviceToHost);
    int diff = 0;
    for (int i = 0; i < n; i++)
        if (a[i] != b[i] && !(x[i] != x[i])) {
            if (diff < 8) std::printf("x=%a (%g) cvt=%04x pk=%04x\n", x[i], x[i], a[i], b[i]);
            diff++;
        }
    std::printf("differences: %d of %d\n", diff, n);
    return 0;
}
