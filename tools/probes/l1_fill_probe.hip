// L1-fill (TCP -> TCC read request) peak of the chip: every lane of every wave reads 4 bytes of its OWN, pseudo-randomly chosen 128-byte line
// of an L2-resident (or MALL / HBM-sized) buffer, 16 reads in flight per lane -- the access shape of a hash-grid gather whose lanes share no
// line.  Prints line fills per second; profile it with --pmc TCP_TCC_READ_REQ_sum to get the counter's unit per fill on this part.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/l1_fill_probe.hip -o /tmp/l1_fill_probe && /tmp/l1_fill_probe [MiB ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256) k_fill(const float* __restrict__ buf, uint32_t line_mask, uint32_t iters, float* __restrict__ out) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; it++) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t line = (x >> 8) & line_mask;
            v[u] = buf[(size_t)line * 32u + (threadIdx.x & 31u)];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) acc += v[u];
    }
    if (acc == 123.456f) out[0] = acc;   // never true: keeps the loads
}

// the other end: every read HITS the CU's L1 (a 16 KB region per workgroup), lanes on pseudo-random dwords of it -- 64 different addresses per
// wave-instruction, i.e. the texture addresser's per-lane rate with no fill behind it
__global__ void __launch_bounds__(256) k_hit(const float* __restrict__ buf, uint32_t iters, float* __restrict__ out) {
    const float* base = buf + (size_t)(blockIdx.x & 255u) * 4096u;      // 16 KB per workgroup
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; it++) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            x = x * 1664525u + 1013904223u;
            v[u] = base[(x >> 8) & 4095u];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) acc += v[u];
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv) {
    std::vector<int> sizes;
    for (int i = 1; i < argc; i++) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) sizes = {2, 16, 49, 512};
    float* out;
    hipMalloc(&out, 4);
    {
        float* hb;
        hipMalloc(&hb, 256 * 16384);
        hipMemset(hb, 0, 256 * 16384);
        const uint32_t blocks = 256 * 8, iters = 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_hit, dim3(blocks), dim3(256), 0, 0, hb, iters, out);
        hipEventRecord(e0, 0);
        const int reps = 5;
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_hit, dim3(blocks), dim3(256), 0, 0, hb, iters, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double loads = (double)blocks * 256 * iters * 16 * reps;
        printf("L1-hit scatter (16 KB per workgroup): %.3f ms per launch, %.3e lane-loads/s\n", ms / reps, loads / (ms * 1e-3));
        hipFree(hb);
    }
    for (int mib : sizes) {
        size_t bytes = (size_t)mib << 20;
        uint32_t lines = 1;
        while ((size_t)lines * 2 * 128 <= bytes) lines *= 2;     // power of two of 128-byte lines
        float* buf;
        hipMalloc(&buf, (size_t)lines * 128);
        hipMemset(buf, 0, (size_t)lines * 128);
        const uint32_t blocks = 256 * 8, iters = 64;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, buf, lines - 1, iters, out);
        hipEventRecord(e0, 0);
        const int reps = 5;
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, buf, lines - 1, iters, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double fills = (double)blocks * 256 * iters * 16 * reps;
        printf("buffer %4zu MiB (%u lines): %.3f ms per launch, %.3e line fills/s, %.2f TB/s of 128-byte lines, %.3e lane-loads per launch\n", ((size_t)lines * 128) >> 20, lines,
               ms / reps, fills / (ms * 1e-3), fills * 128 / (ms * 1e-3) / 1e12, fills / reps);
        hipFree(buf);
    }
    return 0;
}
