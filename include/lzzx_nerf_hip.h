/*
 * lzzx_nerf_hip.h -- C ABI of liblzzx_nerf_hip.so, the gfx950 (MI355X) implementation of the
 * nerf_triplane volumetric-rendering hot path.
 *
 * Drop-in boundary.  Every `lz_*` entry point in sections 1-4 replaces one function of the
 * reference's pybind11 back-ends (file:line cited per entry).  The reference's bindings take
 * at::Tensor; here every tensor is a raw DEVICE pointer to contiguous memory, dimensions are uint32_t,
 * scalars are float, and the last argument is the hipStream_t to launch on (NULL = default stream).
 * Ownership follows the reference: the caller allocates every output; kernels never allocate, free
 * or synchronise.  Return value: 0 on success, otherwise a hipError_t (positive) or LZ_ERR_* (negative);
 * lz_last_error() returns a thread-local message.  Unsupported (D, C) combinations return
 * LZ_ERR_UNSUPPORTED where the reference throws std::runtime_error (gridencoder.cu:354,372).
 *
 * Section 5 holds entry points the reference does not have: the fused per-sample triplane head
 * (the torch-level arithmetic of nerf_triplane/network.py:252-311 as one MFMA kernel) and the
 * device-resident render loop (renderer.py:495-548 without host synchronisation).
 */
#ifndef LZZX_NERF_HIP_H
#define LZZX_NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lz_stream_t; /* hipStream_t */

#define LZ_OK 0
#define LZ_ERR_UNSUPPORTED (-1)
#define LZ_ERR_BAD_ARGUMENT (-2)
/* Conventions.  Every pointer is device memory unless said otherwise; every entry enqueues on `stream` and returns without waiting.  A
 * null pointer where an array is required is LZ_ERR_BAD_ARGUMENT with a message (lz_last_error), never a launch.  A count of ZERO work
 * items (B, N, M, n_alive = 0 with otherwise valid shape parameters; an empty batch of rays, e.g. a rank's empty tile of a small frame)
 * is LZ_OK and launches nothing for every entry whose outputs are per-item arrays, whatever those arrays' pointers are -- an empty
 * torch tensor has no storage (tests/test_cabi.py).  Not per-item, and therefore still required: the loop's `state` / `workspace`
 * words, the parameter blocks, the sums over items of lz_triplane_head_grad_w*.  lz_frame_render with N = 0 still zeroes `state` and,
 * under cap_mode 1, its words of the histogram exchange. */

const char* lz_last_error(void);
/* ABI version of this header; bumped on any signature change */
int lz_abi_version(void);
/* 1 when a gfx950 device is present and usable */
int lz_device_ok(void);

/* ------------------------------------------------------------------------------------------------
 * 1. gridencoder        (reference: gridencoder/src/gridencoder.h:12-13, gridencoder.cu:424-479)
 * ------------------------------------------------------------------------------------------------ */

/* inputs [B,D] f32 in [0,1]; embeddings [sO,C] f32 or f16 (emb_f16); offsets [L+1] i32 (device);
 * outputs: out_layout 0 = [L,B,C] (the reference's level-major layout, gridencoder.cu:95),
 *          out_layout 1 = [B,L*C] (what grid.py:52 produces after its permute; saves that copy); for B >= 65536 the
 *                         buffer is first filled level by level in sample tiles, then untiled in place (same stream);
 *          out_layout 2 = [B,L*C] through the level-resident kernel: a hint that every level's table is <= 64 KB
 *                         (levels that are larger still work, through global gathers, but slowly) -- use for large B;
 * dy_dx [B,L,D,C] or NULL.  S = log2(per_level_scale) as float, H = base resolution. */
int lz_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, void* dy_dx,
                           uint32_t gridtype, int align_corners, int emb_f16, int out_layout, lz_stream_t stream);

/* grad: grad_layout 0 = [L,B,C] (grid.py:70), 1 = [B,L*C]; 2 / 3 = [B,L*C] / [L,B,C] plus the hint that every level's
 * table is <= 64 KB, which selects per-level accumulation in LDS (f32 tables, D <= 3, C <= 2; ignored otherwise; levels
 * that are larger still work, through global atomics); grad_embeddings [sO,C] pre-zeroed by the caller
 * (grid.py:72), accumulated with f32 atomics (f16: packed half2 atomics when C is even, gridencoder.cu:298-304);
 * dy_dx / grad_inputs [B,D] optional (both or neither). */
int lz_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                            void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                            const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners, int emb_f16,
                            int grad_layout, lz_stream_t stream);

/* test hook: flat table index of every corner, [L,B,2^D] i32 (-1 = out of range); exposes get_grid_index
 * (gridencoder.cu:54-72) so index parity can be asserted bit for bit */
int lz_grid_corner_indices(const float* inputs, const int32_t* offsets, int32_t* corner_idx, uint32_t B, uint32_t D,
                           uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                           lz_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 2. shencoder          (reference: shencoder/src/shencoder.h:9-10, shencoder.cu:400-438)
 * ------------------------------------------------------------------------------------------------ */
/* inputs [B,3] f32; outputs [B,degree^2]; dy_dx [B,3,degree^2] or NULL; 1 <= degree <= 8 */
int lz_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree, float* dy_dx,
                         lz_stream_t stream);
/* grad_inputs [B,3] accumulated into (pre-zeroed by the caller, sphere_harmonics.py:49) */
int lz_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t degree,
                          const float* dy_dx, float* grad_inputs, lz_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 3. freqencoder        (reference: freqencoder/src/freqencoder.h:7-10, freqencoder.cu:97-128)
 * ------------------------------------------------------------------------------------------------ */
int lz_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs,
                           lz_stream_t stream);
int lz_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                            float* grad_inputs, lz_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 4. raymarching        (reference: raymarching/src/raymarching.h:7-37)
 * ------------------------------------------------------------------------------------------------ */
int lz_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                          float* nears, float* fars, lz_stream_t stream);                      /* raymarching.h:7  */
int lz_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                    lz_stream_t stream);                                                       /* raymarching.h:8  */
int lz_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, lz_stream_t stream);      /* raymarching.h:9  */
int lz_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, lz_stream_t stream); /* raymarching.h:10 */
int lz_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, lz_stream_t stream); /* :11 */
int lz_morton3D_dilation(const float* grid, uint32_t C, uint32_t H, float* grid_dilation, lz_stream_t stream); /* :12 */

/* raymarching.h:14.  counter [2] i32 = (points, rays), accumulated from its current contents like the
 * reference's atomicAdd; ray rows are emitted in ray-id order (deterministic; one admissible outcome of the
 * reference's unordered atomics, raymarching.cu:446-454).  workspace: >= (N + 2) * 4 bytes of device scratch.
 * Every row of xyzs / dirs / deltas [M, ...] that the call does not write (behind the last sample, in front of the first when the
 * counter did not start at 0, the rows of a ray dropped for lack of room) is set to ZERO by the call itself: the reference's wrapper
 * pre-fills the buffers with torch.zeros (raymarching.py:246-248); here the caller may pass uninitialised memory. */
int lz_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                        uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                        const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                        const float* noises, void* workspace, lz_stream_t stream);
int lz_march_rays_train_backward(const float* grad_xyzs, const float* grad_dirs, const int32_t* rays, const float* deltas,
                                 uint32_t N, uint32_t M, float* grad_rays_o, float* grad_rays_d,
                                 lz_stream_t stream);                                          /* raymarching.h:15 */

/* raymarching.h:19.  xyzs/dirs/deltas pre-zeroed by the caller (raymarching.py:384-386) */
int lz_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                  const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                  const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                  const float* noises, lz_stream_t stream);

/* Compositing, the reference's 13 entry points under their own names and argument order (raymarching.h:16-38) + the stream.
 * Each is a thin wrapper over the three descriptor-driven entries further down (`*_v`), which is what the Python operators call. */
int lz_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ambient, const float* deltas, const int32_t* rays,
                                    uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* ambient_sum, float* depth,
                                    float* image, lz_stream_t stream);                                           /* raymarching.h:16 */
int lz_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_ambient_sum, const float* grad_image,
                                     const float* sigmas, const float* rgbs, const float* ambient, const float* deltas,
                                     const int32_t* rays, const float* weights_sum, const float* ambient_sum, const float* image,
                                     uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas, float* grad_rgbs,
                                     float* grad_ambient, lz_stream_t stream);                                   /* raymarching.h:17 */
int lz_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t, const float* sigmas,
                      const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image,
                      lz_stream_t stream);                                                                       /* raymarching.h:20 */
int lz_composite_rays_ambient(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                              const float* sigmas, const float* rgbs, const float* deltas, const float* ambients, float* weights,
                              float* depth, float* image, float* ambient_sum, lz_stream_t stream);               /* raymarching.h:21 */
int lz_composite_rays_train_sigma_forward(const float* sigmas, const float* rgbs, const float* ambient, const float* deltas,
                                          const int32_t* rays, uint32_t M, uint32_t N, float T_thresh, float* weights_sum,
                                          float* ambient_sum, float* depth, float* image, lz_stream_t stream);   /* raymarching.h:24 */
int lz_composite_rays_train_sigma_backward(const float* grad_weights_sum, const float* grad_ambient_sum, const float* grad_image,
                                           const float* sigmas, const float* rgbs, const float* ambient, const float* deltas,
                                           const int32_t* rays, const float* weights_sum, const float* ambient_sum,
                                           const float* image, uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas,
                                           float* grad_rgbs, float* grad_ambient, lz_stream_t stream);           /* raymarching.h:25 */
int lz_composite_rays_ambient_sigma(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                                    const float* sigmas, const float* rgbs, const float* deltas, const float* ambients,
                                    float* weights, float* depth, float* image, float* ambient_sum, lz_stream_t stream);   /* :27 */
int lz_composite_rays_train_uncertainty_forward(const float* sigmas, const float* rgbs, const float* ambient, const float* uncertainty,
                                                const float* deltas, const int32_t* rays, uint32_t M, uint32_t N, float T_thresh,
                                                float* weights_sum, float* ambient_sum, float* uncertainty_sum, float* depth,
                                                float* image, lz_stream_t stream);                               /* raymarching.h:31 */
int lz_composite_rays_train_uncertainty_backward(const float* grad_weights_sum, const float* grad_ambient_sum,
                                                 const float* grad_uncertainty_sum, const float* grad_image, const float* sigmas,
                                                 const float* rgbs, const float* ambient, const float* uncertainty, const float* deltas,
                                                 const int32_t* rays, const float* weights_sum, const float* ambient_sum,
                                                 const float* uncertainty_sum, const float* image, uint32_t M, uint32_t N,
                                                 float T_thresh, float* grad_sigmas, float* grad_rgbs, float* grad_ambient,
                                                 float* grad_uncertainty, lz_stream_t stream);                   /* raymarching.h:32 */
int lz_composite_rays_uncertainty(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                                  const float* sigmas, const float* rgbs, const float* deltas, const float* ambients,
                                  const float* uncertainties, float* weights, float* depth, float* image, float* ambient_sum,
                                  float* uncertainty_sum, lz_stream_t stream);                                   /* raymarching.h:33 */
int lz_composite_rays_train_triplane_forward(const float* sigmas, const float* rgbs, const float* amb_aud, const float* amb_eye,
                                             const float* uncertainty, const float* deltas, const int32_t* rays, uint32_t M,
                                             uint32_t N, float T_thresh, float* weights_sum, float* amb_aud_sum, float* amb_eye_sum,
                                             float* uncertainty_sum, float* depth, float* image, lz_stream_t stream);    /* :36 */
int lz_composite_rays_train_triplane_backward(const float* grad_weights_sum, const float* grad_amb_aud_sum, const float* grad_amb_eye_sum,
                                              const float* grad_uncertainty_sum, const float* grad_image, const float* sigmas,
                                              const float* rgbs, const float* amb_aud, const float* amb_eye, const float* uncertainty,
                                              const float* deltas, const int32_t* rays, const float* weights_sum,
                                              const float* amb_aud_sum, const float* amb_eye_sum, const float* uncertainty_sum,
                                              const float* image, uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas,
                                              float* grad_rgbs, float* grad_amb_aud, float* grad_amb_eye, float* grad_uncertainty,
                                              lz_stream_t stream);                                               /* raymarching.h:37 */
int lz_composite_rays_triplane(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                               const float* sigmas, const float* rgbs, const float* deltas, const float* ambs_aud,
                               const float* ambs_eye, const float* uncertainties, float* weights, float* depth, float* image,
                               float* amb_aud_sum, float* amb_eye_sum, float* uncertainty_sum, lz_stream_t stream);      /* :38 */

/* The same compositing with the channel variant as data: one entry per direction; the reference's five channel variants
 *   plain (raymarching.h:16-17,20)  ambient (:21)  sigma (:24-27)  uncertainty (:31-33)  triplane (:36-38)
 * are selected by (n_amb, amb_weighted, has_unc) = plain (0,0,0) [inference only], ambient (1,0,0),
 * sigma (1,1,0), uncertainty (1,0,1), triplane (2,0,1).  Unused channel pointers may be NULL.
 * layout: 0 = the reference's ray-major sample rows (the named entries above); 1 = the step-major groups lz_march_rays_train_grouped
 * writes (below).  Per ray the arithmetic and its order are the same under both. */
int lz_composite_train_forward_v(const float* sigmas, const float* rgbs, const float* amb0, const float* amb1,
                                    const float* unc, const float* deltas, const int32_t* rays, uint32_t M, uint32_t N,
                                    float T_thresh, int n_amb, int amb_weighted, int has_unc, int layout, float* weights_sum,
                                    float* amb0_sum, float* amb1_sum, float* unc_sum, float* depth, float* image,
                                    lz_stream_t stream);
/* grad_* outputs pre-zeroed by the caller (raymarching.py:332-334, 649-653) for layout 0; with layout 1 the call writes EVERY row of them
 * (zeros behind a ray's early termination and on rows no ray owns) and the caller may pass uninitialised memory */
int lz_composite_train_backward_v(const float* grad_weights_sum, const float* grad_amb0_sum, const float* grad_amb1_sum,
                                     const float* grad_unc_sum, const float* grad_image, const float* sigmas,
                                     const float* rgbs, const float* amb0, const float* amb1, const float* unc,
                                     const float* deltas, const int32_t* rays, const float* weights_sum,
                                     const float* amb0_sum, const float* unc_sum, const float* image, uint32_t M,
                                     uint32_t N, float T_thresh, int n_amb, int amb_weighted, int has_unc, int layout,
                                     float* grad_sigmas, float* grad_rgbs, float* grad_amb0, float* grad_amb1,
                                     float* grad_unc, lz_stream_t stream);

/* STEP-MAJOR sample rows for training (no reference counterpart: the reference leaves the row order to its atomics, raymarching.cu:446-454,
 * and nothing but rays[] = (ray id, offset, count) says which rows a ray owns).  Rays are taken in `order` (int32 [N], a permutation of
 * 0..N-1; NULL = ray-id order) and GROUPED by G = lz_train_group_size() (a build constant: 16, 32 or 64 lanes of a wave): group g = rays[]
 * rows G g .. G g + G - 1 owns the rows the ray-major layout would give it, ordered by step first:
 *   row(j, k) = o_g + sum_i min(c_i, k) + #{ i < j : c_i > k }   (j = place in the group, k = step, c = counts, o_g = rays[G g].offset).  rays[i] = (order[i], ray-major offset o_i, c_i): the drop rule o_i + c_i > M is the reference's.  A wave of
 * the consumers then holds neighbouring rays at the same step (64 / G steps of G rays) instead of 64 consecutive samples of one ray (head forward of the cfg3
 * step 1.36 -> 0.71 ms).  Consumers: lz_composite_train_{forward,backward}_v with layout = 1, lz_march_rays_train_backward_grouped; the
 * heads and encoders are per-row operators and do not care.  Everything else as lz_march_rays_train (counter, zeroed rows, workspace). */
int lz_march_rays_train_grouped(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                const float* noises, const int32_t* order, void* workspace, lz_stream_t stream);
int lz_train_group_size(void);
int lz_march_rays_train_backward_grouped(const float* grad_xyzs, const float* grad_dirs, const int32_t* rays, const float* deltas,
                                         uint32_t N, uint32_t M, float* grad_rays_o, float* grad_rays_d, lz_stream_t stream);
/* a sort key per ray that puts neighbouring pixels next to each other (direction: octahedral map, 12 + 12 bits Morton-interleaved; origin:
 * 2 bits per axis on top): sort it (stable) to get `order`.  Any permutation is a valid order; this one is the locality heuristic. */
int lz_ray_sort_keys(const float* rays_o, const float* rays_d, uint32_t N, float bound, int32_t* keys, lz_stream_t stream);
/* in place on rays_alive, rays_t and the per-ray accumulators */
int lz_composite_rays_v(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                      const float* sigmas, const float* rgbs, const float* deltas, const float* amb0, const float* amb1,
                      const float* unc, int n_amb, int amb_weighted, int has_unc, float* weights_sum, float* depth,
                      float* image, float* amb0_sum, float* amb1_sum, float* unc_sum, lz_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 5. extensions (no reference counterpart at the FFI level)
 * ------------------------------------------------------------------------------------------------ */

/* ray generation, nerf_triplane/utils.py:226-312: poses device [B,4,4] row-major c2w; rays_o / rays_d [B,N,3].
 * inds: device int64 [N] pixel indices (row * W + col, each in [0, H*W)) shared by the batch, or NULL for every pixel in
 * order (then N must be H * W).  out_i / out_j: optional [B,N] pixel-centre coordinates (results['i'], ['j']). */
int lz_get_rays(const float* poses, float fx, float fy, float cx, float cy, uint32_t H, uint32_t W, uint32_t B, uint32_t N,
                const int64_t* inds, float* rays_o, float* rays_d, float* out_i, float* out_j, lz_stream_t stream);
/* get_bg_coords, nerf_triplane/utils.py:217-223: out [H*W, 2] in [-1, 1] */
int lz_bg_coords(uint32_t H, uint32_t W, float* out, lz_stream_t stream);

/* Fused triplane head: xyz -> 3 x hash-grid (D=2, L=12, C=1) -> aud/eye attention -> sigma net -> SH(4) -> colour net.
 * Weights are consumed in the packed "A-fragment" layout produced by lz_head_pack_weights; all arithmetic is f32
 * (v_mfma_f32_16x16x4_f32), summation order documented in DESIGN.md.  See network.py:252-311. */
#define LZ_HEAD_PACKED_FLOATS 24576 /* 379 A-fragments x 64 lanes + 320 floats of VALU-layer rows */
#define LZ_HEAD_PACKED_F16_BYTES 60416 /* f16 recording forward (training): 59 A-fragments of v_mfma_f32_16x16x32_f16 x 64 lanes x 8 halfs */
#define LZ_HEAD_PACKED_F16W_BYTES 61440 /* f16 inference head: 60 A-fragments of v_mfma_f32_32x32x16_f16 x 64 lanes x 8 halfs */
typedef struct {
    const float* emb_xy;      /* [163584] tables of the three planes (device) */
    const float* emb_yz;
    const float* emb_xz;
    const int32_t* offsets;   /* [13] device */
    const void* packed;       /* packed MLP weights (device), from lz_head_pack_weights (precision 0, 2), lz_head_pack_weights_f16w (precision 1,
                               * inference: lz_triplane_head_forward, lz_frame_*, lz_loop_*) or lz_head_pack_weights_f16 (precision 1, the
                               * recording training forward lz_triplane_head_forward_record_f16) */
    const float* enc_a;       /* [32] device */
    const float* ind_code;    /* [4] device or NULL */
    const float* eye;         /* [1] device or NULL (exp_eye off) */
    float bound;
    float S;                  /* log2(per_level_scale) */
    uint32_t H;               /* base resolution (64) */
    int testing;              /* 1: uncertainty = softplus(0) constant (network.py:243-249) */
    int precision;            /* 0: f32 MFMA, bit-exact fma chains.  1: f16 MFMA with the rounding sequence of the reference
                               * under torch autocast (opt.fp16): half Linear inputs/weights/outputs, f32 accumulate; inference only.
                               * 2: f32 MFMA with geo = Wg s2 folded into color_net.0 -- pack color_net.0 with its geo columns replaced
                               * by W_c0[:, 16:80] . sigma_net.2[1:65] (64 x 64); sigma_net.2's geo rows are then not evaluated (64 of
                               * 361 MFMAs per 16 samples).  Inference only; sigma bit for bit as precision 0, rgb to the reassociation */
} lz_head_params;

/* host-side helper: number of floats lz_head_pack_weights writes */
uint32_t lz_head_packed_size(void);
/* pack the nine bias-free Linear weights (row-major [out,in], DEVICE pointers; unc_* may be NULL) */
int lz_head_pack_weights(const float* aud0, const float* aud1, const float* eye0, const float* eye1, const float* sig0,
                         const float* sig1, const float* sig2, const float* col0, const float* col1, const float* unc0,
                         const float* unc1, int has_eye, int has_ind, float* packed, lz_stream_t stream);
/* the same for the f16 INFERENCE head (no uncertainty net), 32-sample slices on v_mfma_f32_32x32x16_f16 (ABI 10; csrc/lz_head_f16w_slice.h);
 * `packed`: LZ_HEAD_PACKED_F16W_BYTES bytes (device) */
uint32_t lz_head_packed_size_f16w(void);
int lz_head_pack_weights_f16w(const float* aud0, const float* aud1, const float* eye0, const float* eye1, const float* sig0,
                              const float* sig1, const float* sig2, const float* col0, const float* col1, int has_eye,
                              int has_ind, void* packed, lz_stream_t stream);
/* ... and for the f16 recording TRAINING forward (lz_triplane_head_forward_record_f16), 16-sample slices on v_mfma_f32_16x16x32_f16;
 * `packed`: LZ_HEAD_PACKED_F16_BYTES bytes (device) */
uint32_t lz_head_packed_size_f16(void);
int lz_head_pack_weights_f16(const float* aud0, const float* aud1, const float* eye0, const float* eye1, const float* sig0,
                             const float* sig1, const float* sig2, const float* col0, const float* col1, int has_eye,
                             int has_ind, void* packed, lz_stream_t stream);
/* xyzs/dirs [M,3] -> sigmas [M], rgbs [M,3], amb_aud [M], amb_eye [M], unc [M].  `count` (device i32, may be NULL)
 * limits the work to min(M, *count) rows. */
int lz_triplane_head_forward(const lz_head_params* p, const float* xyzs, const float* dirs, uint32_t M,
                             const int32_t* count, float* sigmas, float* rgbs, float* amb_aud, float* amb_eye,
                             float* unc, lz_stream_t stream);

/* Occupancy-grid maintenance, head branch of update_extra_state (nerf_triplane/renderer.py:699-766).
 * lz_density_grid_points: the query point of every cell of every cascade, in meshgrid order (x slowest, z fastest):
 *   xyz = (2 c / (G-1) - 1) * (bound_c - bound_c/G) + (noise * 2 - 1) * bound_c/G, bound_c = min(2^cascade, bound);
 *   noise [C, G^3, 3] in [0,1) is supplied by the caller (the reference draws torch.rand_like per cascade); xyzs [C*G^3, 3].
 * lz_density_grid_update: sigmas [C*G^3] = density at those points (e.g. lz_triplane_head_forward) ->
 *   tmp[cas, morton(cell)] = sigma * density_scale; 6-neighbour dilation (raymarching.cu:304-341); EMA
 *   grid = max(grid * decay, tmp) where both >= 0; stats[0] = mean(clamp(grid, 0)), stats[1] = min(stats[0], density_thresh);
 *   bitfield = packbits(grid, stats[1]).  density_grid [C, G^3] f32 Morton-ordered, in place; bitfield [C*G^3/8];
 *   stats: 2 floats (device); workspace: >= ceil(C*G^3/256) floats (device).  No host synchronisation. */
int lz_density_grid_points(const float* noise, uint32_t C, uint32_t G, float bound, float* xyzs, lz_stream_t stream);
int lz_density_grid_update(const float* sigmas, float density_scale, float decay, float density_thresh, uint32_t C, uint32_t G,
                           float* density_grid, uint8_t* bitfield, float* stats, void* workspace, lz_stream_t stream);

/* mark_untrained_grid (renderer.py:633-695): density_grid[cas, morton(cell)] = -1 for every cell none of the B cameras
 * (poses [B,4,4] c2w, device) sees; count (optional, int32 [C, G^3], Morton-ordered) = cameras covering each cell. */
int lz_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t C, uint32_t G,
                           float bound, float* density_grid, int32_t* count, lz_stream_t stream);

/* Torso half of update_extra_state (renderer.py:772-808).  lz_density_grid_torso_points: query point of every cell of the G x G
 * torso grid in meshgrid order (x slowest), xy = (2 c / (G-1) - 1) * (1 - 1/G) + (noise * 2 - 1) / G; noise [G*G, 2] in [0,1).
 * lz_density_grid_torso_update: alphas [G*G] = forward_torso alpha at those points (lz_torso_forward) -> tmp[y*G + x] -> 5x5 max
 * pool -> density_grid_torso = max(grid * decay, tmp) in place; stats[0] = mean(grid), stats[1] = min(stats[0], density_thresh)
 * (the threshold run_torso masks with, renderer.py:603).  workspace: >= ceil(G*G/256) floats. */
int lz_density_grid_torso_points(const float* noise, uint32_t G, float* xys, lz_stream_t stream);
int lz_density_grid_torso_update(const float* alphas, float decay, float density_thresh, uint32_t G, float* density_grid,
                                 float* stats, void* workspace, lz_stream_t stream);

/* Torso branch of a frame: run_torso's masked query (nerf_triplane/renderer.py:572-631) + forward_torso
 * (nerf_triplane/network.py:170-205) as one kernel, one lane per pixel.  All pointers are device pointers; weights are the
 * reference's bias-free Linear matrices, row-major [out, in], input order [per-pixel features | anchor encoding 42 | ind code]:
 *   deform net  34 + 42 + ind -> 32 -> 32 -> 2,   torso net  32 + 34 + 42 + ind -> 32 -> 32 -> 4.
 * enc_anchor: the frame-constant frequency encoding of the wrapped anchor points (computed by the caller, network.py:179-183). */
typedef struct {
    const float *deform_w0, *deform_w1, *deform_w2;
    const float *torso_w0, *torso_w1, *torso_w2;
    const float* emb;            /* torso_encoder table [sO, 2] (tiledgrid D=2, L=16, C=2, network.py:166) */
    const int32_t* offsets;      /* [17] */
    const float* enc_anchor;     /* [42] */
    const float* ind_code;       /* [ind_dim] or NULL */
    uint32_t ind_dim;            /* 0 or 8 */
    uint32_t gridtype;           /* 0 hash, 1 tiled */
    float torso_shrink;          /* opt.torso_shrink */
    float S;                     /* log2(per_level_scale) of the torso encoder */
    uint32_t H;                  /* its base resolution (16) */
    const float* density_grid;   /* density_grid_torso [G*G] or NULL: no masking */
    uint32_t G;
    float density_thresh;        /* min(density_thresh_torso, mean_density_torso), renderer.py:603 */
} lz_torso_params;
/* enc_anchor [42] of a frame (network.py:179-183): anchor_points [3,4] warped by the inverse of the head pose [4,4] (row-major c2w),
 * (x / w / z, y / w / z) per anchor, frequency-encoded with degree 3 -- one launch instead of torch.inverse's library kernels */
int lz_torso_anchor_encode(const float* pose, const float* anchor_points, float* enc_anchor, lz_stream_t stream);
/* bg_coords [N,2] in [-1,1] -> alpha [N], color [N,3], deform [N,2] (may be NULL); unmasked pixels get zeros */
int lz_torso_forward(const lz_torso_params* p, const float* bg_coords, uint32_t N, float* alpha, float* color, float* deform,
                     lz_stream_t stream);

/* Audio conditioning front-end: NeRFNetwork.encode_audio (nerf_triplane/network.py:226-240) = AudioNet
 * (network.py:40-70) on each of n_win windows a[n_win, dim_in, 16], then, when use_att, AudioAttNet (network.py:9-37) -> enc_a
 * [dim_aud]; without attention enc_a is [n_win, dim_aud].  Weights are the reference's Conv1d [out, in, 3] / Linear [out, in]
 * tensors and biases (device pointers).  One workgroup, one launch (two for wide inputs, see workspace). */
typedef struct {
    const float* c_w[4]; const float* c_b[4];     /* audio_net.encoder_conv.{0,2,4,6}: dim_in->32->32->64->64, k 3, stride 2 */
    const float* fc_w[2]; const float* fc_b[2];   /* audio_net.encoder_fc1.{0,2}: 64->64->dim_aud */
    const float* ac_w[5]; const float* ac_b[5];   /* audio_att_net.attentionConvNet.{0,2,4,6,8}: dim_aud->16->8->4->2->1, k 3 */
    const float* al_w; const float* al_b;         /* audio_att_net.attentionNet.0: Linear(n_win, n_win) */
    uint32_t dim_in, dim_aud, n_win, use_att;
} lz_audio_params;
/* workspace: n_win * 256 floats (device), used when dim_in >= 128 (the first layer then runs as its own chip-wide launch, one
 * wave per output, with a lane-strided summation order); may be NULL for narrower inputs */
int lz_audio_encode(const lz_audio_params* p, const float* a, float* enc_a, void* workspace, lz_stream_t stream);

/* Tall-skinny bias-free Linear for the training path of the heads (the reference's MLP, network.py:73-94, is a stack of
 * nn.Linear(bias=False) with K, N <= 84 over M ~ 1e6..1e7 samples; torch dispatches them to library GEMMs).  Row-major f32,
 * explicit leading dimensions (column slices of wider buffers are fine), v_mfma_f32_16x16x4_f32.
 * forward: Y[M,N] = act(Xm[M,K] . W[N,K]^T), Xm = X where mask > 0 else 0 (mask optional, same layout as X), act = ReLU when
 * relu_out.  The data gradient of a layer is the same call: dX = lz_linear_forward(dY, mask = Y, W^T).  K, N <= 128. */
int lz_linear_forward(const float* X, uint32_t ldx, const float* mask, const float* W, uint32_t ldw, float* Y, uint32_t ldy,
                      uint32_t M, uint32_t K, uint32_t N, int relu_out, lz_stream_t stream);
/* weight gradient: dW[N,K] += (dY where mask > 0)[M,N]^T . X[M,K]; the reduction over M happens inside the kernel (registers ->
 * LDS -> one float atomic per element and workgroup); dW is accumulated into, zero it first.  ceil(N/16)*ceil(K/16) <= 24,
 * K <= 96, N <= 128. */
int lz_linear_grad_w(const float* dY, uint32_t ldd, const float* mask, const float* X, uint32_t ldx, float* dW, uint32_t ldw,
                     uint32_t M, uint32_t K, uint32_t N, lz_stream_t stream);

/* Backward of the fused head for training (f32, testing = 0): recomputes the forward from (xyzs, dirs) and runs the data-gradient
 * chain of NeRFNetwork.forward (network.py:252-311) in one kernel.  Inputs: the upstream gradients of the five head outputs
 * (what composite_rays_train_triplane's backward produces): g_sigma [M], g_rgb [M,3], g_amb_aud [M] (of ||att||), g_amb_eye [M]
 * (of eye_att; may be NULL), g_unc [M].  Outputs (all caller-allocated, row-major f32):
 *   denc   [3,12,M]  d loss / d (plane p's 12 grid features), level-major: denc + p*12*M -> lz_grid_encode_backward(grad_layout 0 / 3)
 *   small  [LZ_BWD_SMALL]  sums over the samples, reduced INSIDE the kernel (registers -> LDS -> one atomic per element and workgroup;
 *          zero it first): d_enc_a [32] | d_ind_code [4] | then the weight gradients of the three skinny output layers
 *          eye_att_net.1 [16] | unc_net.1 [32] | color_net.1 [3,64]
 *   rec    [ceil(M / 16) * 16, LZ_BWD_REC]  one record per sample (16-byte aligned) holding the input X of the wide Linear layers and the gradient G
 *          of their outputs (ReLU mask applied), consumed by lz_triplane_head_grad_w.  One buffer, fixed columns: the kernel needs one
 *          address per sample and immediate offsets (separate buffers cost it two address registers each, spilled), every slot
 *          starts on a 64-byte boundary:
 *            LZ_BWD_X_A1   [64]  input of aud_ch_att_net.1
 *            LZ_BWD_X_SIG0 [69]  input of sigma_net.0; columns 0..35 are enc_x, the input of the three layers stacked in G_X
 *            LZ_BWD_X_S1   [64]  input of sigma_net.1
 *            LZ_BWD_X_S2C  [84]  = [input of sigma_net.2 (s2) 64 | SH(dir) 16 | ind_code 4]: color_net.0's input with geo = s2 . Wg^T
 *                                replaced by s2 itself (Wg = sigma_net.2 rows 1..64)
 *            LZ_BWD_G_X    [112] = [aud_ch_att_net.0 64 | eye_att_net.0 16 (zeros without an eye input) | unc_net.0 32]
 *            LZ_BWD_G_ATT  [32]  aud_ch_att_net.1      LZ_BWD_G_S1 [64]  sigma_net.0      LZ_BWD_G_S2 [64]  sigma_net.1
 *            LZ_BWD_G_C1H  [65]  = [color_net.0 64 | sigma row of sigma_net.2 1]
 *          Neither geo nor d geo is stored: both are linear maps of stored columns, so their weight gradients are finished from the
 *          64 x 64 sum R = G_c1^T s2 (lz_triplane_head_grad_w).  2 624 bytes per sample (3 664 with one buffer per layer). */
/* MEMORY ORDER of rec (and of the state buffer below): blocked by 16-sample slice, [slice][tile of 16 dwords][sample 16][16 dwords] --
 * column c of sample m sits at dword (m / 16) * 16 * ROW + (c / 16) * 256 + (m % 16) * 16 + c % 16 (ROW = dwords per sample) -- so that
 * a wave's store / load instruction (16 samples x 4 lanes x 16 bytes) covers one contiguous kilobyte.  Buffers hold whole slices. */
#define LZ_BWD_REC 656
#define LZ_BWD_SMALL 276
#define LZ_BWD_X_A1 0
#define LZ_BWD_X_SIG0 64
#define LZ_BWD_X_S1 144
#define LZ_BWD_X_S2C 208
#define LZ_BWD_G_X 304
#define LZ_BWD_G_ATT 416
#define LZ_BWD_G_S1 448
#define LZ_BWD_G_S2 512
#define LZ_BWD_G_C1H 576
typedef struct {
    float* denc;
    float* small;
    float* rec;
} lz_head_bwd_out;
int lz_triplane_head_backward(const lz_head_params* p, const float* xyzs, const float* dirs, uint32_t M, const float* g_sigma,
                              const float* g_rgb, const float* g_amb_aud, const float* g_amb_eye, const float* g_unc,
                              const lz_head_bwd_out* out, lz_stream_t stream);
/* The same step without the recompute (lz_head_rec.hip): the forward records, the backward starts from the record.
 * lz_triplane_head_forward_record = lz_triplane_head_forward in training mode (same bits in the five outputs) that also writes the X
 * columns of rec [M, LZ_BWD_REC] (LZ_BWD_X_*: what the recomputing backward wrote itself) and one state row per sample:
 *   state  [ceil(M / 16) * 16, LZ_FWD_STATE] (16-byte aligned, blocked by slice like rec), columns in the lane layout of the kernels (feature 16 t + 4 q + r at 16 t + 4 q + r):
 *            LZ_ST_ATT [32] aud_ch_att_net output   LZ_ST_C1 [64] input of color_net.1   LZ_ST_U1 [32] input of unc_net.1
 *            LZ_ST_E1  [16] input of eye_att_net.1 (unwritten without an eye input)
 *            LZ_ST_MK  [16] four words per q: ReLU masks aud.0 | sigma.0 << 16, sigma.1 | color.0 << 16, unc.0 | eye.0 << 8, and one
 *                           scalar (q = 0 ||att||, 1 eye_att, 2 unc pre-activation, 3 sigma)
 *            LZ_ST_CLR [4]  the three colour pre-activations
 * lz_triplane_head_backward_recorded takes that state instead of (xyzs, dirs): no gather, no forward matrix work (380 instead of 759
 * MFMAs per 16 samples); it fills the G columns of the same rec, denc and small exactly as lz_triplane_head_backward does (same
 * values, same summation order).  The price is memory held from forward to backward: 2 624 + 704 bytes per sample. */
#define LZ_FWD_STATE 176   /* 164 used; rows padded to a multiple of 64 bytes so that every 64-byte store segment covers whole sectors */
#define LZ_ST_ATT 0
#define LZ_ST_C1 32
#define LZ_ST_U1 96
#define LZ_ST_E1 128
#define LZ_ST_MK 144
#define LZ_ST_CLR 160
/* record_f16 = 1: the operands of the weight-gradient products are kept in half precision (rounded to nearest even), which is what the
 * reference's autocast mode feeds its dW GEMMs (opt.fp16: TrainerUtil.py:103, 865-870, with its GradScaler in front); the data-gradient
 * chain and the accumulation stay f32.  rec then holds LZ_BWD_REC16 halves per sample (1 408 bytes instead of 2 624): 16-column tiles
 * interleaved in pairs -- dword j of pair g = {tile 2 g column j, tile 2 g + 1 column j} -- first tile of every slot LZ_R16_*:
 *   X_A1 4 tiles | X_SIG0 6: {enc_x feature 8 (j % 4) + 4 p + j / 4 at tile p column j | feature 32 + q at column 4 q, eye term at
 *   column 1 | enc_a * att 2 tiles | pad} | X_S1 4 | X_S2C 6: {s2 4 | SH component 4 (j % 4) + j / 4 at column j | ind_code[q] at
 *   column 4 q} | G_X 8: {aud.0 4 | eye.0 | unc.0 2 | pad} | G_ATT 2 | G_S1 4 | G_S2 4 | G_C1H 6: {color.0 4 | d h0 at column 0 | pad}
 * and the state row is LZ_FWD_STATE16 dwords: att f32 [32] | c1 2 pairs | u1 1 pair | e1 (low halves) | masks + scalars | colours. */
#define LZ_BWD_REC16 704
#define LZ_R16_X_A1 0
#define LZ_R16_X_SIG0 4
#define LZ_R16_X_S1 10
#define LZ_R16_X_S2C 14
#define LZ_R16_G_X 20
#define LZ_R16_G_ATT 28
#define LZ_R16_G_S1 30
#define LZ_R16_G_S2 34
#define LZ_R16_G_C1H 38
#define LZ_FWD_STATE16 128
#define LZ_S16_C1 32
#define LZ_S16_U1 64
#define LZ_S16_E1 80
#define LZ_S16_MK 96
#define LZ_S16_CLR 112
int lz_triplane_head_forward_record(const lz_head_params* p, const float* xyzs, const float* dirs, uint32_t M, float* sigmas, float* rgbs,
                                    float* amb_aud, float* amb_eye, float* unc, void* rec, float* state, int record_f16, lz_stream_t stream);
int lz_triplane_head_backward_recorded(const lz_head_params* p, const float* state, uint32_t M, const float* g_sigma, const float* g_rgb,
                                       const float* g_amb_aud, const float* g_amb_eye, const float* g_unc, const lz_head_bwd_out* out,
                                       int record_f16, const void* packed_bwd16, lz_stream_t stream);
/* packed_bwd16 (optional, with record_f16 = 1): the transposed weights as half fragments; the data-gradient products dX = W^T dY then
 * run on v_mfma_f32_16x16x16_f16 with dY and W rounded to half and an f32 sum -- the arithmetic of the reference's autocast backward
 * (half Linear gradients, TrainerUtil.py:865-870) -- one instruction per (input tile, output tile) instead of four f32 ones.
 * eye0 may be NULL without an eye input.  lz_head_packed_bwd_size_f16() bytes, 16-byte aligned. */
uint32_t lz_head_packed_bwd_size_f16(void);
int lz_head_pack_weights_bwd_f16(const float* aud0, const float* aud1, const float* eye0, const float* sig0, const float* sig1,
                                 const float* sig2, const float* col0, int has_eye, int has_ind, void* packed_bwd16, lz_stream_t stream);
/* The recording forward on the f16 matrix cores (lz_head_rec16.hip): the forward of the reference's usual training mode (autocast,
 * TrainerUtil.py:865; rounding sequence of the f16 inference head, plus the uncertainty net) writing the f16 records and state row
 * described above -- sigma / rgb / ambient outputs have the bits of lz_triplane_head_forward(precision 1), unc = softplus (f32) of the
 * half pre-activation.  p->packed: lz_head_pack_weights_f16 image, p->precision = 1, p->testing = 0; packed_unc: the five fragments of
 * unc_net from lz_head_pack_unc_f16 (lz_head_packed_unc_size_f16() bytes, 16-byte aligned).  The backward is
 * lz_triplane_head_backward_recorded(record_f16 = 1) with the f32 image of the same weights (data gradient in f32), the weight
 * gradients lz_triplane_head_grad_w_f16. */
uint32_t lz_head_packed_unc_size_f16(void);
int lz_head_pack_unc_f16(const float* unc0, const float* unc1, void* packed_unc, lz_stream_t stream);
int lz_triplane_head_forward_record_f16(const lz_head_params* p, const void* packed_unc, const float* xyzs, const float* dirs, uint32_t M,
                                        float* sigmas, float* rgbs, float* amb_aud, float* amb_eye, float* unc, void* rec16, float* state16,
                                        lz_stream_t stream);
/* inputs of the three table scatters of a training step in one launch: out [3, M, 2] = (x, y) | (y, z) | (x, z) of xyzs [M, 3], each mapped
 * (v + bound) / (2 bound) exactly as the fused forward maps it (grid.py:143; network.py:208-223 for the plane order) */
int lz_triplane_plane_coords(const float* xyzs, uint32_t M, float bound, float* out, lz_stream_t stream);
/* Weight gradients of the wide layers from the records, in ONE pass over them (five waves per workgroup, each owning the
 * accumulator tiles of one product; partial tiles per workgroup in `workspace`, summed by a second small launch: no atomics).
 * Outputs are overwritten, row-major [N, K]: dW_x3 [112,36] = aud_ch_att_net.0 (rows 0..63) | eye_att_net.0 (64..79) | unc_net.0
 * (80..111); dW_aud1 [32,64]; dW_sig0 [64,k_sig0] (k_sig0 = 69 with the eye column, else 68); dW_sig1 [64,64]; dW_c1h [65,84] =
 * [G_c1 | d h0]^T . [s2 | SH | ind]: with R = rows 0..63 x columns 0..63, Wg = sigma_net.2.weight[1:65] and Wc = color_net.0.weight,
 *   color_net.0.weight.grad = [dW_c1h[0:64, 64:80] | R . Wg^T | dW_c1h[0:64, 80:84]]
 *   sigma_net.2.weight.grad = [dW_c1h[64, 0:64] ; Wc[:, 16:80]^T . R]
 * (two 64^3 products, left to the caller).  workspace: lz_triplane_head_grad_w_workspace() bytes of device memory. */
size_t lz_triplane_head_grad_w_workspace(void);
int lz_triplane_head_grad_w(const float* rec, uint32_t M, uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0,
                            float* dW_sig1, float* dW_c1h, void* workspace, lz_stream_t stream);
/* the same pass over f16 records (LZ_BWD_REC16 halves per sample, see lz_triplane_head_forward_record): operands converted to f32 at
 * use, f32 accumulation, same outputs */
int lz_triplane_head_grad_w_f16(const void* rec16, uint32_t M, uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0,
                                float* dW_sig1, float* dW_c1h, void* workspace, lz_stream_t stream);
/* The backward over f16 records with the weight gradients of the wide layers reduced INSIDE the kernel: =
 * lz_triplane_head_backward_recorded(record_f16 = 1, packed_bwd16) followed by lz_triplane_head_grad_w_f16, without the G half of the
 * records ever being written or the records read a second time.  The waves of a workgroup transpose their G / X tiles through LDS
 * (ds_read_b64_tr_b16) and share the 95 accumulator tiles.  packed_bwd16 non-null: data gradient on the f16 matrix cores (with the f16
 * recording forward: the whole step in the reference's `-O` arithmetic); NULL: the f32 data-gradient chain through p->packed.  rec16: the
 * f16 records of lz_triplane_head_forward_record(record_f16 = 1) / _forward_record_f16 (read only: the X half); out->rec is ignored;
 * same operand rounding as the two-pass arrangement, f32 accumulation in a different (fixed) order; color_net.1's weight gradient is one
 * of the products here (its dY rounded to half like the others). */
int lz_triplane_head_backward_recorded_dw16(const lz_head_params* p, const float* state, const void* rec16, uint32_t M, const float* g_sigma,
                                            const float* g_rgb, const float* g_amb_aud, const float* g_amb_eye, const float* g_unc,
                                            const lz_head_bwd_out* out, const void* packed_bwd16, uint32_t k_sig0, float* dW_x3,
                                            float* dW_aud1, float* dW_sig0, float* dW_sig1, float* dW_c1h, void* workspace,
                                            lz_stream_t stream);
/* The same `-O` step with the MLP RECOMPUTED in the backward (ABI 10): nothing travels from the forward to the backward but the enc_x
 * operand the MLP started from -- encx16: [ceil(M / 16)][5][64] dwords, LZ_ENCX16_BYTES(M) = 80 bytes per sample instead of the 1 216 of
 * f16 record + state -- and the view directions.  lz_triplane_head_forward_encx_f16 = lz_triplane_head_forward_record_f16 (same arithmetic,
 * same five outputs, bit for bit) writing encx16 only; lz_triplane_head_backward_encx_dw16 = lz_triplane_head_backward_recorded_dw16 whose
 * waves first run the f16 forward chain of their slice again (packed_f16: the lz_head_pack_weights_f16 image the forward used; packed_unc:
 * lz_head_pack_unc_f16) and keep the layer inputs, state pairs and ReLU masks in registers; packed_bwd16 is required (data gradient on the
 * f16 matrix cores).  Same gradients as the recorded pair, bit for bit.  p->packed: the f32 image (lz_head_pack_weights), training mode. */
#define LZ_ENCX16_BYTES(M) ((size_t)(((M) + 15u) / 16u) * 1280u)
int lz_triplane_head_forward_encx_f16(const lz_head_params* p, const void* packed_unc, const float* xyzs, const float* dirs, uint32_t M,
                                      float* sigmas, float* rgbs, float* amb_aud, float* amb_eye, float* unc, void* encx16, lz_stream_t stream);
int lz_triplane_head_backward_encx_dw16(const lz_head_params* p, const void* packed_f16, const void* packed_unc, const void* encx16,
                                        const float* dirs, uint32_t M, const float* g_sigma, const float* g_rgb, const float* g_amb_aud,
                                        const float* g_amb_eye, const float* g_unc, const lz_head_bwd_out* out, const void* packed_bwd16,
                                        uint32_t k_sig0, float* dW_x3, float* dW_aud1, float* dW_sig0, float* dW_sig1, float* dW_c1h,
                                        void* workspace, lz_stream_t stream);

/* Device-resident inference loop (renderer.py:495-548): no host synchronisation inside the frame, 3 launches per iteration:
 *     lz_loop_march -> lz_triplane_head_forward(count = state words + LZ_LOOP_NEXT + 2) -> lz_loop_composite.
 * The device buffer passed as `lz_loop_state*` must hold LZ_LOOP_STATE_INTS int32:
 *   words  0..7   the struct below: the state as of the last COMMITTED iteration (what the host inspects);
 *   words  8..71  per-workgroup sample-count slots of the march (folded into total_samples at commit);
 *   word   72     LZ_LOOP_STAT_ROWS: sample rows handed to the head so far (sum of n_alive * n_step, exhausted rows included);
 *   words 74..79  LZ_LOOP_NEXT: {n_alive, n_step, n_samples, step, done, iterations} of the iteration in flight.
 * lz_loop_march advances the state itself: every workgroup sums the per-workgroup survivor counts the previous compositing
 * launch left in `workspace` (its own prefix = compaction offset, and the total = n_alive), applies the schedule rule, and
 * workgroup 0 publishes the "next" record; lz_loop_composite reads that record and its workgroup 0 commits it to the struct.
 * Once `done` is set every later launch is a no-op; the host must enqueue one iteration after the last real one for the
 * struct to show done = 1. */
#define LZ_LOOP_STATE_INTS 80
#define LZ_LOOP_STAT_ROWS 72
#define LZ_LOOP_NEXT 74
typedef struct {
    int32_t n_alive;      /* rays alive in the last committed iteration = length of the list the next march compacts */
    int32_t n_step;       /* max(min(sample_budget / n_alive, n_step_cap), 1); budget = N, cap = 8: renderer.py:513 */
    int32_t step;         /* sum of n_step before that iteration */
    int32_t done;         /* 1 once n_alive == 0 or step >= max_steps */
    int32_t n_samples;    /* n_alive * n_step of that iteration */
    int32_t total_samples;/* marched samples so far (delta != 0) */
    int32_t iterations;   /* iterations executed (-1 right after lz_loop_begin) */
    int32_t pad;
} lz_loop_state;

/* `workspace`: >= 4096 int32 of device scratch (per-workgroup survivor counts); at most 4096 * 256 rays per frame.  The alive
 * list ping-pongs between two [N] int32 buffers. */

/* Iteration schedule: the reference marches n_step = max(min(N / n_alive, 8), 1) samples per alive ray per iteration
 * (renderer.py:513), i.e. a budget of N sample rows per iteration and at most 8 steps.  `sample_budget` / `n_step_cap`
 * generalise the two constants (0 = the reference's N / 8): a larger budget means fewer, fatter iterations.  Pixels do
 * not depend on the schedule (rays are independent and compositing resumes exactly); per-ray marched-sample counts do
 * only for rays that terminate early inside a chunk.  Sample buffers must hold max(sample_budget, N) rows. */

/* rays_alive <- 0..N-1, rays_t <- nears, accumulators <- 0, state <- pre-state (list of N survivors, no steps taken),
 * workspace <- workgroup sizes */
/* start of every ray with `perturb` (renderer.py:344,521 -> raymarching.cu:873, first iteration only): t0 = near + clamp(near * dt_gamma,
 * dt_min, dt_max) * noise, evaluated as the reference's fused multiply-add; hand t0 to lz_loop_begin in place of `nears` */
int lz_perturb_starts(const float* nears, const float* noises, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, uint32_t N,
                      float* t0, lz_stream_t stream);
int lz_loop_begin(uint32_t N, uint32_t max_steps, uint32_t sample_budget, uint32_t n_step_cap, const float* nears,
                  int32_t* rays_alive, float* rays_t,
                  float* weights_sum, float* depth, float* image, float* amb0_sum, float* amb1_sum, float* unc_sum,
                  lz_loop_state* state, void* workspace, lz_stream_t stream);
/* state advance (see above) + order-preserving stream compaction of rays_alive_in (drops the -1 entries compositing left,
 * renderer.py:542) into rays_alive_out + march of the survivors; writes zero rows for exhausted rays; adds the marched sample
 * count to the slots and, when ray_counts != NULL, per ray to ray_counts[ray id] */
int lz_loop_march(lz_loop_state* state, uint32_t N, uint32_t sample_budget, uint32_t n_step_cap, const int32_t* rays_alive_in,
                  int32_t* rays_alive_out, const void* workspace, const float* rays_t, const float* rays_o,
                  const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                  const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                  int32_t* ray_counts, lz_stream_t stream);
/* triplane compositing on the current list (in place), per-workgroup survivor counts into workspace, commit of the state */
int lz_loop_composite(lz_loop_state* state, uint32_t N, float T_thresh, int32_t* rays_alive, float* rays_t,
                      const float* sigmas, const float* rgbs, const float* deltas, const float* amb0, const float* amb1,
                      const float* unc, float* weights_sum, float* depth, float* image, float* amb0_sum, float* amb1_sum,
                      float* unc_sum, void* workspace, lz_stream_t stream);
/* lz_loop_composite for a network without ambient / uncertainty channels (composite_rays, raymarching.h:17) */
int lz_loop_composite_plain(lz_loop_state* state, uint32_t N, float T_thresh, int32_t* rays_alive, float* rays_t,
                            const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                            float* image, void* workspace, lz_stream_t stream);
/* diagnostic (synchronises the device): out2[0] = shader-clock cycles, out2[1] = 100 MHz wall-clock ticks that wave 0 of
 * workgroup 0 spent inside the most recent lz_triplane_head_forward launch; ratio x 100 MHz = sustained shader clock */
int lz_debug_head_clocks(uint64_t* out2);

/* Everything one frame's loop touches (all device pointers; sample buffers hold max(sample_budget, N) rows since
 * n_alive * n_step <= max(sample_budget, N)). */
typedef struct {
    lz_head_params head;
    lz_loop_state* state;          /* LZ_LOOP_STATE_INTS int32 */
    void* workspace;               /* >= 4096 int32 */
    int32_t* rays_alive[2];        /* ping-pong alive lists, [N] each */
    float* rays_t;                 /* [N] */
    const float* rays_o;           /* [N,3] */
    const float* rays_d;           /* [N,3] */
    const float* nears;            /* [N] */
    const float* fars;             /* [N] */
    const uint8_t* grid;           /* density bitfield */
    float* xyzs;  float* dirs;  float* deltas;                             /* [N,3] [N,3] [N,2] */
    float* sigmas;  float* rgbs;  float* amb_aud;  float* amb_eye;  float* unc;   /* [N] [N,3] [N] [N] [N] */
    float* weights_sum;  float* depth;  float* image;                     /* [N] [N] [N,3] */
    float* amb_aud_sum;  float* amb_eye_sum;  float* unc_sum;             /* [N] each */
    int32_t* ray_counts;           /* [N] or NULL */
    uint32_t N, max_steps, C, H;
    float bound, dt_gamma, T_thresh;
    uint32_t sample_budget, n_step_cap;   /* 0 = the reference's schedule (N, 8) */
} lz_frame;

/* enqueue `n_iterations` iterations (march -> head -> composite -> advance) back to back; `parity` = index of the
 * alive list that is current before the first of them (0 after lz_loop_begin, then the count of iterations run so far
 * modulo 2).  Iterations past the end of the frame are no-ops on the device. */
typedef struct lz_timing lz_timing;   /* opaque: pairs of HIP events */
int lz_timing_create(uint32_t n_pairs, lz_timing** out);
int lz_timing_destroy(lz_timing* t);
int lz_timing_reset(lz_timing* t);
/* record the begin (which = 0) / end (which = 1) event of the next pair on `stream` */
int lz_timing_mark(lz_timing* t, int which, lz_stream_t stream);
/* elapsed ms of every recorded pair (host array); call after synchronising the stream */
int lz_timing_elapsed_ms(lz_timing* t, float* out_ms, uint32_t capacity, uint32_t* n_pairs);
/* `timing` (may be NULL): bracket every head launch with an event pair on the launch stream */
int lz_loop_run(const lz_frame* f, uint32_t parity, uint32_t n_iterations, lz_timing* timing, lz_stream_t stream);

/* One inference frame as ONE persistent kernel (csrc/lz_frame.hip): near/far + first occupied cell + longest-rays-first queue
 * (2 small launches), then march -> head -> composite per ray slot with refill from the queue; no per-sample buffers, no host round
 * trip.  Pixels, depth and sums of a ray do not depend on the iteration schedule except through the CAP: the reference's loop
 * (renderer.py:406-570) tests `step < max_steps` once per iteration and advances step += n_step, n_step = max(min(N // n_alive, 8), 1)
 * (:503,513,546), so every ray still alive at the cap has received the same frame-wide count C_eff = sum of n_step, in [max_steps,
 * max_steps + 7].
 *   cap_mode 1 (LZ_FRAME_CAP_REFERENCE): reproduces that.  Phase 1 stops rays at exactly max_steps and leaves, per ray, the last chunk
 *     boundary it can survive; a histogram of those replays n_alive / n_step / C_eff on the device (no host round trip); phase 2
 *     continues the rays that stood at the cap to C_eff samples.  Equal to the reference loop pixel for pixel whatever steps_per_pass is;
 *     with ray_counts also count for count (a ray cut by T_thresh inside a chunk counts the chunk's marched samples, raymarching.py:347-398).
 *     Ranks that render tiles of ONE frame set N_total and defer_finish, all-reduce (sum) cap_ws[0 .. max_steps] after lz_frame_render and
 *     then call lz_frame_finish: their tiles equal the unsharded reference frame.  Needs ray_last, cap_ws; max_steps <= 4096.
 *   cap_mode 0 (LZ_FRAME_CAP_PER_RAY): a ray alive at the cap stops at ceil(max_steps / S) * S samples, S = steps_per_pass: the loop under
 *     the schedule n_step = S.  No histogram, no second phase, no exchange between ranks -- and not the reference's pixels on rays that
 *     reach the cap (every ray that leaves the box or falls under T_thresh before max_steps is unaffected).
 * All pointers are device pointers; the caller owns every buffer.  state words after the call (stream order):
 *   [1] rays that had at least one sample, [3] 1, [5] composited samples (with ray_counts under cap_mode 1: marched samples = the sum of
 *   ray_counts), [6] 1, [72] sample rows evaluated by the head (16 per slice) -- words 3 / 5 / 6 / 72 as in lz_loop_state /
 *   LZ_LOOP_STAT_ROWS; cap_mode 1 adds [9] rays phase 2 continued, [10] C_eff, [11] iterations of the reference's loop. */
#define LZ_FRAME_CAP_PER_RAY 0
#define LZ_FRAME_CAP_REFERENCE 1
#define LZ_FRAME_CAP_WS_INTS(max_steps) (2 * (max_steps) + 24)
#define LZ_FRAME_STATE_INTS 1024
typedef struct {
    lz_head_params head;           /* testing = 1; precision 0 (f32) or 1 (f16) */
    const float* rays_o;           /* [N,3] */
    const float* rays_d;           /* [N,3] */
    const uint8_t* grid;           /* density bitfield */
    const float* aabb;             /* [6] */
    float* nears;  float* fars;    /* [N] each, written (near_far_from_aabb) */
    float* rays_t;                 /* [N] scratch: t of every ray's first occupied cell */
    int32_t* order;                /* [N] scratch: the queue */
    int32_t* state;                /* LZ_FRAME_STATE_INTS, zeroed by the call */
    uint8_t* keys;                 /* [N] scratch */
    float* weights_sum;  float* depth;  float* image;                     /* [N] [N] [N,3] */
    float* amb_aud_sum;  float* amb_eye_sum;  float* unc_sum;             /* [N] each */
    float* out;                    /* [N,3] clamp(image + (1 - weights_sum) * bg, 0, 1), renderer.py:559-561 */
    const float* bg;               /* [N,3] or NULL -> bg_scalar */
    uint8_t* out_rgb24;            /* [N,3] or NULL: (out * 255) truncated, TrainerUtil.py:550-555 */
    int32_t* ray_counts;           /* [N] or NULL: samples per ray */
    float bg_scalar, bound, dt_gamma, T_thresh, min_near;
    uint32_t N, max_steps, C, H;
    uint32_t steps_per_pass;       /* samples a ray marches per pass = the n_step of the schedule it equals: 0 = auto (1 for large
                                      frames; 2..16 when there are too few rays to fill the chip), or 1, 2, 4, 8, 16 */
    const float* noises;           /* [N] or NULL: `perturb` of the reference's inference loop (renderer.py:521 passes it on the first
                                      iteration only): every ray starts at near + clamp(near * dt_gamma, dt_min, dt_max) * noise
                                      (raymarching.cu:873) */
    const float* occupied_aabb;    /* [6] or NULL: lz_occupied_bounds of `grid`.  The march is then confined to that box: the stretch in
                                      front of it is walked with the march's own step t += clamp(t * dt_gamma, dt_min, dt_max) and no cell
                                      test, and a ray ends where it leaves the box -- the same samples (no cell outside the box is
                                      occupied, and the t sequence of a ray does not depend on what the cells hold), without the ~130
                                      instructions per empty cell crossed */
    float* t_end;                  /* [N] scratch, required with occupied_aabb: where each ray's march ends */
    uint32_t cap_mode;             /* LZ_FRAME_CAP_PER_RAY | LZ_FRAME_CAP_REFERENCE (see above) */
    uint32_t N_total;              /* cap_mode 1: rays of the whole frame this call renders a tile of = the N of renderer.py:513 (0 = N) */
    uint32_t defer_finish;         /* cap_mode 1: 1 = lz_frame_render stops behind the histogram; the caller (all-reduces it and) calls
                                      lz_frame_finish with the same struct */
    int32_t* ray_last;             /* cap_mode 1: [N] scratch */
    int32_t* cap_ws;               /* cap_mode 1: LZ_FRAME_CAP_WS_INTS(max_steps) int32, zeroed by lz_frame_render; words [0 .. max_steps] =
                                      histogram over the rays of the last chunk boundary each can survive (bin max_steps: alive at the cap) */
} lz_frame_fused;
/* World-space bounds {xmin, ymin, zmin, xmax, ymax, zmax} of the occupied cells of a density bitfield (bit index = level * H^3 +
 * morton(x, y, z), raymarching.cu:267-300), every level's cells dilated by `margin` cells of their own size and by at least four
 * of the march's longest steps (4 dt_max, raymarching.cu:866); a side that reaches the rim
 * of the outermost level is open (-/+FLT_MAX: the march clamps positions to the bound before the cell test); an empty bitfield gives a box
 * at +FLT_MAX that no ray reaches.  workspace: 48 int32.  Two small launches, no host round trip; run it when the bitfield changes. */
int lz_occupied_bounds(const uint8_t* bitfield, uint32_t C, uint32_t H, float bound, uint32_t margin, int32_t* workspace,
                       float* aabb6, lz_stream_t stream);
struct lz_timing;
/* `timing` (may be NULL): bracket the PHASE-1 persistent launch with one event pair on the launch stream (lz_timing_create).  Under
 * cap_mode 1 the histogram / schedule kernels, the phase-2 persistent launch (rays parked at max_steps) and lz_k_frame_counts run behind
 * the pair and are NOT in it: where the cap binds (max_steps 16, small tiles) the pair understates the frame's kernel time -- time the
 * whole call (bench.py's ms_per_step) or take the kernel trace there; where it does not bind those launches return at once (~10 us) */
int lz_frame_render(const lz_frame_fused* f, struct lz_timing* timing, lz_stream_t stream);
/* cap_mode 1 with defer_finish: schedule replay from cap_ws, phase 2, marched counts (see above); lz_frame_render calls it itself otherwise */
int lz_frame_finish(const lz_frame_fused* f, lz_stream_t stream);

/* ---- BASELINE cfg2: a generic hash-grid NeRF on the operators of encoding.get_encoder (encoding.py:6-37) ---------------------------
 * hashgrid (input_dim 3, num_levels 16, level_dim 2; gridencoder.h:12) -> sigma MLP 32-64-16 -> SH(4) + 15 geometry features -> colour
 * MLP 31-64-3, bias-free Linear + ReLU (network.py:73-94), sigma = exp(row 0), rgb = sigmoid.
 *
 * lz_grid_encode_forward_tiled: the level-major gather of lz_grid_encode_forward alone, leaving the TILED layout
 *     outputs = [tile][level][sample in tile][C], tiles of LZ_GRID_TILE_ROWS samples, a last partial tile of n rows as [level][n][C]
 * for consumers that read it in place (the head below), so the [B, L*C] matrix is never untiled.  bound > 0: inputs arrive in
 * [-bound, bound] and are mapped (x + bound) / (2 bound) like GridEncoder.forward (grid.py:143).  count (device int32, may be NULL):
 * rows in use this launch -- tiles behind it are skipped. */
#define LZ_GRID_TILE_ROWS 256
int lz_grid_encode_forward_tiled(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B,
                                 const int32_t* count, float bound, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                 uint32_t gridtype, int align_corners, int emb_f16, lz_stream_t stream);
/* The per-sample network behind the gather as ONE kernel on v_mfma_f32_16x16x4_f32 (96 MFMAs per 16 samples).  packed: LZ_NGP_FRAGS
 * fragments of 64 floats (lzzx_nerf_amd/ngp.py: pack_weights; fragment (ks, ft), lane l = W[16 ft + (l & 15)][k(ks, l >> 4)], zero padded);
 * feats: feat_layout 0 = f32 [rows, 32] row-major (lz_grid_encode_forward, out_layout 1), 1 = the tiled f32 layout above, 2 = tiled f16
 * (half tables: features are widened to f32, as `.float()` does in front of the reference's Linear); dirs [rows, 3]; count as above;
 * sigmas [rows], rgbs [rows, 3]. */
#define LZ_NGP_FRAGS 96
int lz_ngp_head_forward(const float* packed, const void* feats, int feat_layout, const float* dirs, uint32_t rows, const int32_t* count,
                        float* sigmas, float* rgbs, lz_stream_t stream);
/* Everything one frame of the hash-grid NeRF loop touches (device pointers; sample buffers hold max(sample_budget, N) rows). */
typedef struct {
    const float* packed;           /* LZ_NGP_FRAGS * 64 floats */
    const void* embeddings;        /* hash table, f32 or f16 [offsets[L], 2] */
    const int32_t* offsets;        /* [L + 1] */
    uint32_t enc_L, enc_H;         /* num_levels, base_resolution */
    float enc_S;                   /* log2(per_level_scale), as gridencoder.h:12 */
    int32_t emb_f16;
    void* feats;                   /* rows * 32 elements of the table's type (tiled features) */
    lz_loop_state* state;          /* LZ_LOOP_STATE_INTS int32 */
    void* workspace;               /* >= 4096 int32 */
    int32_t* rays_alive[2];
    float* rays_t;
    const float* rays_o;  const float* rays_d;  const float* nears;  const float* fars;
    const uint8_t* grid;           /* density bitfield */
    float* xyzs;  float* dirs;  float* deltas;  float* sigmas;  float* rgbs;
    float* weights_sum;  float* depth;  float* image;
    int32_t* ray_counts;           /* [N] or NULL */
    uint32_t N, max_steps, C, H;
    float bound, dt_gamma, T_thresh;
    uint32_t sample_budget, n_step_cap;   /* 0 = the reference's schedule (N, 8) */
} lz_frame_ngp;
/* enqueue `n_iterations` iterations (march -> gather -> head -> composite, 4 launches each) back to back; parity as lz_loop_run */
int lz_ngp_loop_run(const lz_frame_ngp* f, uint32_t parity, uint32_t n_iterations, lz_stream_t stream);

/* Multi-GPU tile hand-off without a collective (lzzx_nerf_amd/dist.py: PeerTileGatherer): every rank copies its rendered tile straight
 * into each peer's frame buffer (one xGMI hop), then raises its flag there; lz_wait_flags makes `stream` wait, ON THE DEVICE, until all `n`
 * flags (int32, written by the peers' copies) have reached `want` -- a BOUNDED spin (`max_polls` polls of ~1 us each): on expiry it sets
 * *timed_out = 1 and returns, so a dead peer costs a wrong frame and an error flag, never a hung GPU. */
int lz_wait_flags(const int32_t* flags, uint32_t n, int32_t want, uint32_t max_polls, int32_t* timed_out, lz_stream_t stream);

/* image = clamp(image + (1 - weights_sum) * bg, 0, 1) (renderer.py:559-561); bg: device [N,3] or NULL -> bg_scalar */
int lz_final_blend(const float* image, const float* weights_sum, const float* bg, float bg_scalar, uint32_t N,
                   float* out, lz_stream_t stream);
/* the same plus the hand-off format of the reference's video pipe, (pred * 255).astype(np.uint8) (TrainerUtil.py:550-555,
 * 668): out_rgb24 [N,3] u8, truncating; `out` (f32) may be NULL */
int lz_final_blend_rgb24(const float* image, const float* weights_sum, const float* bg, float bg_scalar, uint32_t N,
                         float* out, uint8_t* out_rgb24, lz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LZZX_NERF_HIP_H */
