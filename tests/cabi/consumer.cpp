// A consumer of the C ABI that knows nothing about Python or torch: plain hipMalloc'd buffers in, results out, every check against values
// worked out on the host here.  Built and run by tests/test_gpu_cabi_consumer.py:
//   hipcc -Iinclude tests/cabi/consumer.cpp -Llzzx_nerf_amd/lib -llzzx_nerf_hip -Wl,-rpath,<lib dir> -o consumer
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lzzx_nerf_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("hip: %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define LZ_OK_(x) do { int r_ = (x); if (r_ != 0) { std::printf("lz: %d %s (%s:%d)\n", r_, lz_last_error(), __FILE__, __LINE__); return 3; } } while (0)
#define CHECK(c) do { if (!(c)) { std::printf("check failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 4; } } while (0)

static uint32_t spread3(uint32_t v) {   // raymarching.cu:56-63
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

int main() {
    CHECK(lz_abi_version() >= 9);
    CHECK(lz_device_ok() == 1);
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    // ---- morton3D / morton3D_invert round trip over a cube of coordinates
    const uint32_t R = 24, N = R * R * R;
    std::vector<int32_t> coords(N * 3), idx(N), back(N * 3);
    for (uint32_t i = 0; i < N; i++) { coords[3 * i] = i % R; coords[3 * i + 1] = (i / R) % R; coords[3 * i + 2] = i / (R * R); }
    int32_t *d_coords, *d_idx, *d_back;
    HIP_OK(hipMalloc(&d_coords, N * 12)); HIP_OK(hipMalloc(&d_idx, N * 4)); HIP_OK(hipMalloc(&d_back, N * 12));
    HIP_OK(hipMemcpyAsync(d_coords, coords.data(), N * 12, hipMemcpyHostToDevice, st));
    LZ_OK_(lz_morton3D(d_coords, N, d_idx, st));
    LZ_OK_(lz_morton3D_invert(d_idx, N, d_back, st));
    HIP_OK(hipMemcpyAsync(idx.data(), d_idx, N * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(back.data(), d_back, N * 12, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < N; i++) {
        const uint32_t want = spread3(coords[3 * i]) | (spread3(coords[3 * i + 1]) << 1) | (spread3(coords[3 * i + 2]) << 2);
        CHECK((uint32_t)idx[i] == want);
        CHECK(back[3 * i] == coords[3 * i] && back[3 * i + 1] == coords[3 * i + 1] && back[3 * i + 2] == coords[3 * i + 2]);
    }
    // ---- packbits: bit i of byte b <-> grid[8 b + i] > thresh (raymarching.cu:267-300)
    const uint32_t G = 4096;
    std::vector<float> grid(G);
    for (uint32_t i = 0; i < G; i++) grid[i] = (float)((i * 2654435761u) >> 24) / 255.0f;
    std::vector<uint8_t> bits(G / 8);
    float* d_grid; uint8_t* d_bits;
    HIP_OK(hipMalloc(&d_grid, G * 4)); HIP_OK(hipMalloc(&d_bits, G / 8));
    HIP_OK(hipMemcpyAsync(d_grid, grid.data(), G * 4, hipMemcpyHostToDevice, st));
    LZ_OK_(lz_packbits(d_grid, G, 0.5f, d_bits, st));
    HIP_OK(hipMemcpyAsync(bits.data(), d_bits, G / 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (uint32_t b = 0; b < G / 8; b++) {
        uint8_t want = 0;
        for (int i = 0; i < 8; i++) want |= (grid[8 * b + i] > 0.5f) ? (1u << i) : 0u;
        CHECK(bits[b] == want);
    }
    // ---- near_far_from_aabb on rays whose slab intersection is known in closed form: origin (0, 0, -3), direction through the box centre
    const uint32_t NR = 3;
    const float ro[NR * 3] = {0, 0, -3, 0, 0, -3, 5, 5, -3};
    const float rd[NR * 3] = {1e-3f, 1e-3f, 1.0f, 1e-3f, 1e-3f, 1.0f, 1e-3f, 1e-3f, 1.0f};   // third ray misses the box
    const float aabb[6] = {-1, -1, -1, 1, 1, 1};
    float *d_ro, *d_rd, *d_aabb, *d_near, *d_far, near[NR], far[NR];
    HIP_OK(hipMalloc(&d_ro, sizeof(ro))); HIP_OK(hipMalloc(&d_rd, sizeof(rd))); HIP_OK(hipMalloc(&d_aabb, sizeof(aabb)));
    HIP_OK(hipMalloc(&d_near, sizeof(near))); HIP_OK(hipMalloc(&d_far, sizeof(far)));
    HIP_OK(hipMemcpyAsync(d_ro, ro, sizeof(ro), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_rd, rd, sizeof(rd), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_aabb, aabb, sizeof(aabb), hipMemcpyHostToDevice, st));
    LZ_OK_(lz_near_far_from_aabb(d_ro, d_rd, d_aabb, NR, 0.05f, d_near, d_far, st));
    HIP_OK(hipMemcpyAsync(near, d_near, sizeof(near), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(far, d_far, sizeof(far), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    CHECK(std::fabs(near[0] - 2.0f) < 1e-5f && std::fabs(far[0] - 4.0f) < 1e-5f);
    CHECK(near[1] == near[0] && far[1] == far[0]);
    CHECK(near[2] > 1e30f && far[2] > 1e30f);   // a miss is FLT_MAX on both (raymarching.cu:120-145)
    // ---- argument errors come back as codes with a message, never as a crash
    CHECK(lz_morton3D(nullptr, 4, d_idx, st) != 0 && lz_last_error()[0] != 0);
    // ... on every family of entry points (round 2's incident: a null pointer launched instead of rejected faulted the GPU)
    float* fnull = nullptr;
    CHECK(lz_grid_encode_forward(nullptr, nullptr, nullptr, nullptr, 4, 2, 1, 1, 1.0f, 16, nullptr, 0, 0, 0, 0, st) != 0);
    CHECK(lz_grid_encode_backward(nullptr, nullptr, nullptr, nullptr, nullptr, 4, 2, 1, 1, 1.0f, 16, nullptr, nullptr, 0, 0, 0, 0, st) != 0);
    CHECK(lz_grid_corner_indices(nullptr, nullptr, nullptr, 4, 2, 1, 1, 1.0f, 16, 0, 0, st) != 0);
    CHECK(lz_sh_encode_forward(nullptr, nullptr, 4, 3, 4, nullptr, st) != 0);
    CHECK(lz_sh_encode_backward(nullptr, nullptr, 4, 3, 4, nullptr, nullptr, st) != 0);
    CHECK(lz_freq_encode_forward(nullptr, 4, 2, 4, 18, nullptr, st) != 0);
    CHECK(lz_freq_encode_backward(nullptr, nullptr, 4, 2, 4, 18, nullptr, st) != 0);
    CHECK(lz_triplane_head_forward(nullptr, fnull, fnull, 16, nullptr, fnull, fnull, fnull, fnull, fnull, st) != 0);
    lz_head_params hp{};                                  // zeroed parameter block: every table / weight pointer null
    CHECK(lz_triplane_head_forward(&hp, d_near, d_near, 16, nullptr, d_near, d_near, d_near, d_near, d_near, st) != 0);
    CHECK(lz_march_rays_train(nullptr, nullptr, nullptr, 1.0f, 0.0f, 16, 4, 1, 128, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                              nullptr, d_idx, st) != 0);
    CHECK(lz_composite_rays_train_triplane_forward(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 8, 4, 1e-4f, nullptr, nullptr, nullptr,
                                                   nullptr, nullptr, nullptr, st) != 0);
    CHECK(lz_near_far_from_aabb(nullptr, nullptr, nullptr, 4, 0.05f, nullptr, nullptr, st) != 0);
    CHECK(lz_packbits(nullptr, 8, 0.5f, nullptr, st) != 0);
    CHECK(lz_occupied_bounds(nullptr, 1, 128, 1.0f, 8, nullptr, nullptr, st) != 0);
    CHECK(lz_occupied_bounds(d_bits, 9, 128, 1.0f, 8, d_idx, d_near, st) != 0);          // cascade out of range
    CHECK(lz_torso_anchor_encode(nullptr, nullptr, nullptr, st) != 0);
    CHECK(lz_frame_render(nullptr, nullptr, st) != 0);
    // ... and the entry points of ABI 9: the deferred half of the reference's cap, the plain compositing of the loop, the hash-grid NeRF path
    CHECK(lz_frame_finish(nullptr, st) != 0);
    lz_frame_fused ff{};                                  // zeroed frame: no rays, no buffers
    ff.cap_mode = LZ_FRAME_CAP_REFERENCE;
    CHECK(lz_frame_render(&ff, nullptr, st) != 0);
    CHECK(lz_frame_finish(&ff, st) != 0);
    CHECK(lz_loop_composite_plain(nullptr, 4, 1e-4f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st) != 0);
    CHECK(lz_grid_encode_forward_tiled(nullptr, nullptr, nullptr, nullptr, 256, nullptr, 1.0f, 3, 2, 16, 1.0f, 16, 0, 0, 0, st) != 0);
    CHECK(lz_grid_encode_forward_tiled(d_near, d_near, d_idx, d_near, 256, nullptr, 1.0f, 2, 2, 16, 1.0f, 16, 0, 0, 0, st) != 0);   // D = 3, C = 2 only
    CHECK(lz_ngp_head_forward(nullptr, nullptr, 0, nullptr, 16, nullptr, nullptr, nullptr, st) != 0);
    CHECK(lz_ngp_head_forward(d_near, d_near, 7, d_near, 16, nullptr, d_near, d_near, st) != 0);                                       // unknown layout
    CHECK(lz_ngp_loop_run(nullptr, 0, 1, st) != 0);
    lz_frame_ngp fn{};
    CHECK(lz_ngp_loop_run(&fn, 0, 1, st) != 0);
    CHECK(lz_last_error()[0] != 0);
    HIP_OK(hipStreamSynchronize(st));                     // nothing was launched: the stream is still healthy
    LZ_OK_(lz_packbits(d_grid, G, 0.5f, d_bits, st));
    HIP_OK(hipStreamSynchronize(st));
    std::printf("cabi consumer ok\n");
    return 0;
}
