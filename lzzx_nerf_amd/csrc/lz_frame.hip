// lz_frame.hip -- one inference frame as ONE persistent kernel: march -> triplane gather -> MLP (MFMA) -> composite per ray, with
// the per-sample buffers of the reference loop (xyzs, dirs, deltas, sigmas, rgbs, ambient, uncertainty: ~60 B per sample through HBM)
// gone and ~90 launches per frame (3 per iteration of nerf_triplane/renderer.py:503-548) replaced by 3.
//
// What it computes is the reference's inference loop (renderer.py:406-570) under the iteration schedule n_step = 1: every alive ray
// marches one sample per iteration, the head evaluates it, compositing resumes the ray.  The schedule is a launch-shape heuristic
// (rays are independent, compositing resumes exactly), so pixels, depth, ambient / uncertainty sums and -- except for rays cut by
// T_thresh inside a multi-step chunk -- per-ray sample counts equal those of any other schedule; the arithmetic of every step is the
// operator kernels' (lz_march.h: LzMarch::probe; lz_head_slice.h / lz_head_f16_slice.h; lz_k_composite_rays), bit for bit.
//
// MI355X design
//   * A wave owns 16 RAY SLOTS = the 16 samples of one MFMA B-operand tile.  Per pass: every slot marches its ray to the next
//     occupied cell (lanes q == 0; the DDA of LzMarch), the 64 lanes evaluate the head for the 16 samples exactly like a slice of
//     lz_k_triplane_head, lanes q == 0 composite the result into the slot's accumulators.  Ray state (t, far, 8 accumulators, count)
//     lives in LDS between passes (12 KB per workgroup), not in registers: the head's register budget is unchanged.
//   * Slots that finish (ray left the box, T < T_thresh, max_steps) are refilled from a global queue with one wave-aggregated atomic,
//     so a slice never carries exhausted rows: the "ray compaction" of renderer.py:542 is the refill.
//   * Queue order = longest rays first (counting sort by the estimated sample count (far - t_first) / dt, 256 bins): the rays handed
//     out last are the short ones, which bounds the tail where slots run empty (list scheduling, LPT).
//   * lz_k_frame_prepare (one lane per ray) does near/far, marches every ray to its FIRST occupied cell -- rays that never meet
//     one get their background pixel there and never enter the queue -- and builds the histogram; lz_k_frame_scatter places the
//     ray ids; lz_k_frame is the persistent kernel: 3 launches + 1 memset per frame, no host round trip, no per-sample HBM traffic
//     (compulsory: 24 B/ray in, ~50 B/ray out).
//   * No inter-workgroup communication; a wave leaves when the queue is dry and its slots are empty, so the grid always drains.
//
// The cap (cap_mode 1, the default of the Python renderer).  The reference tests `step < max_steps` once per ITERATION and advances
// step += n_step with n_step = max(min(N // n_alive, 8), 1) (renderer.py:503-548), so every ray still alive at the cap has received the
// same GLOBAL count C_eff = sum of n_step, somewhere in [max_steps, max_steps + 7] -- a property of the whole frame, not of the ray.
// A ray is alive at the iteration that starts at sample boundary B exactly when B <= L, L = min(samples the box holds, tau - 1), tau =
// the sample at which T < T_thresh fired (compositing kills a ray whose chunk was cut short, raymarching.cu:2193-2238).  So:
//     phase 1   the persistent kernel as above with the cap at exactly max_steps; every ray leaves its L in ray_last (max_steps =
//               still alive), rays at the cap park their accumulators in the output arrays and their t in rays_t;
//     lz_k_frame_cap_hist   histogram of L over the rays (LDS per workgroup), parked rays compacted into the queue;
//     [ranks rendering tiles of ONE frame all-reduce the max_steps + 1 histogram words here: lz_frame_finish]
//     lz_k_frame_schedule   replays n_alive / n_step from the histogram: C_eff, the chunk boundaries;
//     phase 2   the persistent kernel again over the parked rays, from their parked state, up to C_eff samples (no work, one early-out
//               launch, when no ray reached the cap -- the 192-step headline frame);
//     lz_k_frame_counts (only with ray_counts)   a ray cut by T_thresh inside a chunk was still MARCHED to the end of that chunk by the
//               reference (raymarching.py:347-398): its count grows by the box samples behind tau up to the chunk's end.
// cap_mode 0 keeps the per-ray cap ceil(max_steps / S) * S (no histogram, no second phase).
#include <stdlib.h>

#include "lz_march.h"
#include "lz_head_slice.h"
#include "lz_head_f16_slice.h"
#include "lz_head_f16w_slice.h"

#ifndef LZF_WG
#define LZF_WG 1024   // (-DLZF_WG=768: three waves per SIMD, 168 registers; round 5 tried the f16 slice with ONE 72-load gather there: 1.84 ms against 1.77, and 2.33 at 1024 threads)
#endif
#define LZF_WAVES (LZF_WG / 64)
#define LZF_BINS 256
#define LZF_LUT 256          // Morton bit-spread table in LDS, for grids up to 256^3 (the reference hard-codes 128, renderer.py:94)
#define LZF_MARCH_PROBES 2   // empty cells a slot may cross per march attempt (S = 1); measured on cfg5 (f16 / f32 ms): 1 -> 1.80 / 4.30, 2 -> 1.82 / 4.29, 3 -> 1.83 / 4.34, 6 -> 1.84 / 4.38, 12 -> 1.96 / 4.49, unbounded -> 2.39 / 4.81
// device state words (LZ_FRAME_STATE_INTS int32, zeroed per frame by lz_frame_render).  Words 3, 5, 6 and 72 sit where the
// multi-launch loop keeps done / total_samples / iterations / rows (lz_loop_state, LZ_LOOP_STAT_ROWS), so a caller reads both alike.
#define LZF_Q_HEAD 0      // queue cursor
#define LZF_Q_SIZE 1      // rays in the queue (those with at least one sample)
#define LZF_DONE 3        // 1 (the frame is complete in stream order)
#define LZF_SAMPLES 5     // marched = composited samples
#define LZF_ITER 6        // 1: one persistent launch
#define LZF_ROWS 72       // sample rows handed to the head (16 per slice)
// cap_mode 1 (the reference's cap, see "the cap" below)
#define LZF_P_HEAD 8      // phase 2: queue cursor over the rays phase 1 parked at max_steps
#define LZF_P_SIZE 9      // rays parked by phase 1 that phase 2 continues (0 when C_eff == max_steps)
#define LZF_CEFF 10       // C_eff: samples a ray alive at the cap receives under the reference's schedule
#define LZF_SCHED_K 11    // iterations the reference's loop runs
#define LZF_TICKET 12     // workgroups of lz_k_frame_cap_hist that have flushed
#define LZF_CAP_MAX_STEPS 4096   // LDS histogram / schedule tables of the cap kernels
#define LZF_HIST 128      // [256] rays per key
#define LZF_CURSOR 384    // [256] scatter cursors

struct LzFrameK {
    const float* rays_o; const float* rays_d; const uint8_t* grid; const float* aabb;
    float* nears; float* fars; float* rays_t;
    int* order; int* state; uint8_t* keys;
    float* weights_sum; float* depth; float* image; float* amb0_sum; float* amb1_sum; float* unc_sum;
    float* out; const float* bg; uint8_t* out_rgb24; int* ray_counts;
    float bg_scalar, bound, dt_gamma, T_thresh, min_near;
    uint32_t N, max_steps, C, H;
    const float* noises;
    const float* occ;     // [6] or null: bounds of the occupied cells (lz_occupied_bounds); the march is confined to them
    float* t_end;         // [N] with occ: where a ray's march ends (far, clipped to occ); the persistent kernel reads it instead of fars
    // ---- the cap (cap_mode 1) ----
    int* ray_last;        // [N]: L of every ray = the last chunk boundary it can survive, min(box samples, tau - 1), max_steps = parked at the cap
    int* cap_ws;          // [0 .. max_steps] histogram of L (what ranks all-reduce), then the schedule tables (lz_k_frame_schedule)
    uint32_t cap_mode, phase2, N_total;
    LzMarchFrame mf;      // the march's frame-wide quotients (lz_march_frame on the host)
};

// The per-ray OUTPUT side of LzFrameK (eleven pointers, the background, the cap's buffers) is touched once per ray, when it leaves its slot.
// Read as ordinary kernel arguments these fields are loop-invariant scalars, the compiler keeps all of them in scalar registers through every
// pass, and the f32 frame kernel -- short of scalar registers next to the head's -- spills them to vector lanes and restores them inside the
// pass loop (round 4 measured that: ten more live scalars = +0.9 % on the headline frame).  LzfOut reads a field from the kernel-argument
// segment AT THE POINT OF USE instead (a volatile scalar load: not hoisted), for the price of a few s_load per finished ray.
typedef const char __attribute__((address_space(4))) lz_kernarg_t;
struct LzfOut {
    lz_kernarg_t* f;      // where the kernel's LzFrameK argument sits in its kernel-argument segment
    template <typename T> __device__ __forceinline__ T get(size_t off) const {
        return *reinterpret_cast<const volatile T __attribute__((address_space(4)))*>(f + off);
    }
};
#define LZF_OUT(o, name) ((o).template get<decltype(LzFrameK::name)>(__builtin_offsetof(LzFrameK, name)))
// the hand-computed offset assumes that LzFrameK is laid out in the kernel-argument segment like any 8-byte-aligned trivially copyable struct
// argument, directly behind ARGS_BEFORE bytes of arguments rounded up to 8 (lz_k_frame_prep: first argument; lz_k_frame: behind HD::Args)
static_assert(alignof(LzFrameK) == 8 && __is_trivially_copyable(LzFrameK), "LZF_OUT reads LzFrameK from the kernel-argument segment at round8(ARGS_BEFORE)");
template <size_t ARGS_BEFORE>    // bytes of kernel arguments in front of the LzFrameK (0: it is the first)
__device__ __forceinline__ LzfOut lzf_out() {
    return LzfOut{(lz_kernarg_t*)__builtin_amdgcn_kernarg_segment_ptr() + ((ARGS_BEFORE + 7) & ~size_t(7))};
}

__device__ __forceinline__ void lzf_write_pixel(const LzfOut& O, int ray, float ws, float d, float r, float g, float b, float a0, float a1,
                                                float u, int cnt) {
    LZF_OUT(O, weights_sum)[ray] = ws;
    LZF_OUT(O, depth)[ray] = d;
    const float rgb[3] = {r, g, b};
    float* image = LZF_OUT(O, image);
    float* out = LZF_OUT(O, out);
    const float* bg = LZF_OUT(O, bg);
    uint8_t* out_rgb24 = LZF_OUT(O, out_rgb24);
    const float bg_scalar = LZF_OUT(O, bg_scalar);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const size_t t = (size_t)ray * 3 + c;
        image[t] = rgb[c];
        const float bgv = bg ? bg[t] : bg_scalar;
        const float v = rgb[c] + (1.0f - ws) * bgv;              // renderer.py:559, two roundings like lz_k_final_blend
        const float cl = lz_fminf(lz_fmaxf(v, 0.0f), 1.0f);
        out[t] = cl;
        if (out_rgb24) out_rgb24[t] = (uint8_t)(cl * 255.0f);
    }
    LZF_OUT(O, amb0_sum)[ray] = a0;
    LZF_OUT(O, amb1_sum)[ray] = a1;
    LZF_OUT(O, unc_sum)[ray] = u;
    int* ray_counts = LZF_OUT(O, ray_counts);
    if (ray_counts) ray_counts[ray] = cnt;
}

// ---- pass 1: near / far, first occupied cell, sort key, histogram -------------------------------------------------------------
// (1 024 rays per workgroup in both small passes: each workgroup ends with one global atomic per non-empty key, and same-address atomics
// serialise in the L2 at ~80 ns each -- with 256-ray workgroups a 512^2 frame queued ~1 000 of them on every popular key)
#define LZF_PREP_WG 1024
__global__ void __launch_bounds__(LZF_PREP_WG) lz_k_frame_prepare(LzFrameK F) {
    __shared__ int hist[LZF_BINS];
    __shared__ uint32_t mlut[LZF_LUT];                       // Morton bit-spread table for the march (as in lz_k_frame)
    if (F.cap_mode && blockIdx.x == 0)                       // the cap's histogram and tables: first touched two launches later
        for (uint32_t i = threadIdx.x; i < LZ_FRAME_CAP_WS_INTS(F.max_steps); i += blockDim.x) F.cap_ws[i] = 0;
    if (threadIdx.x < LZF_BINS) {
        hist[threadIdx.x] = 0;
        mlut[threadIdx.x] = lz_expand_bits(threadIdx.x);
    }
    __syncthreads();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < F.N) {
        const float* o = F.rays_o + (size_t)n * 3;
        const float* d = F.rays_d + (size_t)n * 3;
        float near, far;
        lz_near_far_ray(o[0], o[1], o[2], d[0], d[1], d[2], F.aabb, F.min_near, near, far);
        F.nears[n] = near;
        F.fars[n] = far;
        LzMarch m;
        m.init(o, d, F.bound, F.dt_gamma, F.max_steps, F.C, F.H, F.grid);
        if (F.H <= LZF_LUT) m.morton_lut = mlut;
        float t = near, x, y, z, dt = 0.0f;
        if (F.noises) t = lz_fmaf(lz_clampf(t * F.dt_gamma, m.dt_min, m.dt_max), F.noises[n], t);   // perturb: raymarching.cu:873, first iteration only
        if (F.occ) {
            // No cell outside the (dilated) bounds of the occupied ones can yield a sample, and every step of the reference's march --
            // the sample step and the empty-cell skip alike -- is t += clamp(t * dt_gamma, dt_min, dt_max) (raymarching.cu:907, 919-926):
            // the t sequence of a ray does not depend on what the cells hold.  So the stretch in front of the bounds is walked with that
            // one addition per step instead of a cell test per step (~130 instructions), and the march ends where the ray leaves them.
            float on, of;
            lz_near_far_ray(o[0], o[1], o[2], d[0], d[1], d[2], F.occ, near, on, of);
            if (on == on && of == of) {          // (a NaN from an axis-parallel ray: no clipping)
                const float stop = lz_fminf(on, far);
                while (t < stop) t += lz_clampf(t * F.dt_gamma, m.dt_min, m.dt_max);
                far = lz_fminf(far, of);
            }
            F.t_end[n] = far;
        }
        bool found = false;
        while (t < far) {
            if (m.probe(t, x, y, z, dt)) { found = true; break; }
        }
        int key = 0;
        if (found && F.max_steps > 0) {
            F.rays_t[n] = t;
            // upper estimate of the samples left on the ray: steps of at least the current dt until far
            const float est = lz_fminf((far - t) / dt + 1.0f, (float)F.max_steps);
            key = (int)(est * 255.0f / (float)F.max_steps);
            key = key < 1 ? 1 : (key > 255 ? 255 : key);
#ifdef LZF_KEY_MERGE      /* experiment: 2^LZF_KEY_MERGE neighbouring length bins share a queue stretch (order inside: ray order) */
            key |= (1 << LZF_KEY_MERGE) - 1;
#endif
            atomicAdd(&hist[key], 1);
        } else {
            lzf_write_pixel(lzf_out<0>(), (int)n, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0);   // no sample on this ray: background
            if (F.cap_mode) F.ray_last[n] = 0;     // alive in the reference's first iteration only
        }
        F.keys[n] = (uint8_t)key;
    }
    __syncthreads();
    if (threadIdx.x > 0 && threadIdx.x < LZF_BINS) {
        const int h = hist[threadIdx.x];
        if (h) atomicAdd(F.state + LZF_HIST + threadIdx.x, h);
    }
}

// ---- pass 2: ray ids in descending key order (counting sort; order inside a bin is free -- rays are independent) ---------------
// Per workgroup: a local histogram in LDS gives every ray its rank among the workgroup's rays of the same key (one LDS atomic each),
// ONE global atomic per non-empty key reserves the workgroup's range in that bin, all of them in flight together.
__global__ void __launch_bounds__(LZF_PREP_WG) lz_k_frame_scatter(LzFrameK F) {
    __shared__ int start[LZF_BINS], lcount[LZF_BINS], lbase[LZF_BINS];
    const bool bin = threadIdx.x < LZF_BINS;                  // the first 256 lanes also own a key each
    {
        // start[k] = rays with a larger key (they come first): suffix scan of the 256 bins (key 0 = no sample, not queued)
        const int h = (bin && threadIdx.x > 0) ? F.state[LZF_HIST + threadIdx.x] : 0;
        if (bin) { start[threadIdx.x] = h; lcount[threadIdx.x] = 0; }
        __syncthreads();
        for (int off = 1; off < LZF_BINS; off <<= 1) {
            const int v = (bin && threadIdx.x + off < LZF_BINS) ? start[threadIdx.x + off] : 0;
            __syncthreads();
            if (bin) start[threadIdx.x] += v;
            __syncthreads();
        }
        const int incl = bin ? start[threadIdx.x] : 0;      // rays with key >= k
        __syncthreads();
        if (bin) start[threadIdx.x] = incl - h;
        if (blockIdx.x == 0 && threadIdx.x == 1) { F.state[LZF_Q_SIZE] = incl; F.state[LZF_DONE] = 1; F.state[LZF_ITER] = 1; }   // keys 1..255
    }
    __syncthreads();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    const int key = n < F.N ? (int)F.keys[n] : 0;
    const int lrank = key > 0 ? atomicAdd(&lcount[key], 1) : 0;
    __syncthreads();
    if (bin && threadIdx.x > 0) {
        const int c = lcount[threadIdx.x];
        if (c > 0) lbase[threadIdx.x] = atomicAdd(F.state + LZF_CURSOR + threadIdx.x, c);
    }
    __syncthreads();
    if (key > 0) F.order[start[key] + lbase[key] + lrank] = (int)n;
}

// ---- pass 3: the persistent kernel --------------------------------------------------------------------------------------------
// slot state in LDS, per wave [field][16]
enum { SF_RAY = 0, SF_T, SF_FAR, SF_DT, SF_WS, SF_D, SF_R, SF_G, SF_B, SF_A0, SF_A1, SF_U, SF_CNT,
       SF_SH };                     // SH(4) of the ray's direction, evaluated once per ray at refill (the head needs it per sample): 16 f32
                                    // words, or 8 words of packed halves for the f16 head.  The staging fields behind it depend on the
                                    // kernel's arrangement: LzfFields
// fields behind SF_SH.  S > 1: per-SAMPLE staging (march -> head: X Y Z TS; head -> composite: OSIG .. OU) and the per-ray pass counter IT;
// S == 1 with several slot rows: only the parked head outputs; one row: nothing
template <int PREC, int S, int ROWS> struct LzfFields {
    static constexpr int SH_WORDS = PREC == 1 ? 8 : 16;
    static constexpr int ST = SF_SH + SH_WORDS;
    static constexpr int X = ST, Y = ST + 1, Z = ST + 2, TS = ST + 3;                       // S > 1 only
    static constexpr int OSIG = S > 1 ? ST + 4 : ST, OR = OSIG + 1, OG = OSIG + 2, OB = OSIG + 3, OA0 = OSIG + 4, OA1 = OSIG + 5, OU = OSIG + 6;
    static constexpr int IT = OSIG + 7;                                                       // S > 1 only
    static constexpr int RD = S > 1 ? ST + 12 : (ROWS > 1 ? ST + 7 : ST);                    // 1 / direction (3 words), set when the slot takes the ray
    // origin and direction of the slot's ray (6 words) behind the reciprocals: LzMarch::init reads LDS instead of two dwordx3 global loads at the
    // head of every pass's dependent chain (same-box A/B, round 4: f32 frame 8.95 -> 8.90 ms, f16 1.990 -> 1.985, f16 8-way tile 0.433 ->
    // 0.426).  Not with three slot rows: the f16 kernel's LDS is full there (157 KB).
    static constexpr bool GEO = ROWS < 3;
    static constexpr int COUNT = RD + 3 + (GEO ? 6 : 0);
};

// SH(4) of the slot's ray from LDS: component k of the ray whose state sits at slot `ls` (fields are `ns` slots wide)
struct LzShFromSlot {
    const float* slot;
    int ls, ns;
    __device__ __forceinline__ void prepare() const {}
    __device__ __forceinline__ float comp_iq(int i, int q) const { return slot[(SF_SH + 4 * i + q) * ns + ls]; }
    __device__ __forceinline__ float comp_qj(int q, int j) const { return slot[(SF_SH + 4 * q + j) * ns + ls]; }
};
// the f16 head consumes SH as halves: the slot keeps them packed (components 2 k, 2 k + 1 in word k of 8), converted once per ray
struct LzShFromSlot16 {
    const float* slot;
    int ls, ns;
    __device__ __forceinline__ void prepare() const {}
};
__device__ __forceinline__ void w_sh_pk(const LzShFromSlot16& f, int h, uint32_t (&w)[4]) {    // components 8 h .. 8 h + 7: words 4 h .. 4 h + 3
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = __float_as_uint(f.slot[(SF_SH + 4 * h + k) * f.ns + f.ls]);
}
// evaluated by the lane that takes the ray (the same lz_sh_eval call on the same direction as the stand-alone head makes per sample)
template <int PREC, bool GEO = false>
__device__ __forceinline__ void lzf_store_sh(const float* __restrict__ ro, const float* __restrict__ d, float* slot, int s, int ns, int rd_field) {
    float o[16];
    lz_sh_eval(d[0], d[1], d[2], 4, o, nullptr, nullptr, nullptr);
    slot[rd_field * ns + s] = 1 / d[0];              // LzMarch's reciprocals: per ray here, not per pass
    slot[(rd_field + 1) * ns + s] = 1 / d[1];
    slot[(rd_field + 2) * ns + s] = 1 / d[2];
    if constexpr (GEO) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            slot[(rd_field + 3 + k) * ns + s] = ro[k];
            slot[(rd_field + 6 + k) * ns + s] = d[k];
        }
    }
    if constexpr (PREC == 1) {
#pragma unroll
        for (int k = 0; k < 8; k++) slot[(SF_SH + k) * ns + s] = __uint_as_float(h_cvt2(o[2 * k], o[2 * k + 1], false));
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) slot[(SF_SH + k) * ns + s] = o[k];
    }
}

// a slot takes a ray: fresh accumulators (phase 1), or the ones phase 1 parked in the output arrays when the ray reached max_steps (phase 2)
__device__ __forceinline__ void lzf_slot_take(const LzfOut& O, bool ph2, int ray, float* slot, int* sloti, int sl, int ns) {
    if (__builtin_expect(ph2, 0)) {
        slot[SF_WS * ns + sl] = LZF_OUT(O, weights_sum)[ray];
        slot[SF_D * ns + sl] = LZF_OUT(O, depth)[ray];
        const float* image = LZF_OUT(O, image);
        slot[SF_R * ns + sl] = image[(size_t)ray * 3];
        slot[SF_G * ns + sl] = image[(size_t)ray * 3 + 1];
        slot[SF_B * ns + sl] = image[(size_t)ray * 3 + 2];
        slot[SF_A0 * ns + sl] = LZF_OUT(O, amb0_sum)[ray];
        slot[SF_A1 * ns + sl] = LZF_OUT(O, amb1_sum)[ray];
        slot[SF_U * ns + sl] = LZF_OUT(O, unc_sum)[ray];
        sloti[SF_CNT * ns + sl] = (int)LZF_OUT(O, max_steps);
    } else {
#pragma unroll
        for (int f = SF_WS; f <= SF_U; f++) slot[f * ns + sl] = 0.0f;
        sloti[SF_CNT * ns + sl] = 0;
    }
}
// a ray leaves its slot.  kind: how it ended -- LZF_END_BOX no further sample in the box (`composited` = all it has), LZF_END_T the
// compositing cut it at its `composited`-th sample (T < T_thresh), LZF_END_CAP alive after `composited` = cap samples.  cap_mode 1
// records L for the schedule (phase 1), parks a capped ray's t for phase 2, and flags the count of a T-cut ray (negative, its t kept)
// for lz_k_frame_counts; `report` is the count written otherwise.
enum { LZF_END_BOX = 0, LZF_END_T = 1, LZF_END_CAP = 2 };
__device__ __forceinline__ void lzf_ray_end(const LzfOut& O, bool ph2, int ray, int kind, int composited, int report, float t, float ws, float d,
                                            float r, float g, float b, float a0, float a1, float u) {
    if (__builtin_expect(LZF_OUT(O, cap_mode) != 0, 0)) {
        report = composited;
        if (kind == LZF_END_T) {
            if (LZF_OUT(O, ray_counts)) { report = -composited; LZF_OUT(O, rays_t)[ray] = t; }
            if (!ph2) LZF_OUT(O, ray_last)[ray] = composited - 1;
        } else if (!ph2) {
            LZF_OUT(O, ray_last)[ray] = composited;          // LZF_END_CAP: composited == max_steps, the bin of the rays phase 2 continues
            if (kind == LZF_END_CAP) LZF_OUT(O, rays_t)[ray] = t;
        }
    }
    lzf_write_pixel(O, ray, ws, d, r, g, b, a0, a1, u, report);
}

template <int PREC> struct LzfHead;
template <> struct LzfHead<0> {
    using Args = LzHeadArgs; using Ctx = LzHeadCtx; using Out = LzHeadOut; using ShSlot = LzShFromSlot;
    static constexpr int LDS_WORDS = LzHeadLds<false>::FLOATS;
    __device__ static __forceinline__ void stage(const Args& P, float* lds, int q, Ctx& c) { lz_head_stage<false>(P, lds, LZF_WG, q, c); }
    template <typename ShFn>
    __device__ static __forceinline__ void slice(const Ctx& c, int lane, float x, float y, float z, ShFn f, Out& o) {
        lz_head_slice<false, false, true, true>(c, lane, x, y, z, f, o);
    }
};
template <> struct LzfHead<2> : LzfHead<0> {   // f32 with the geo projection folded into colour_net.0 (lz_head_slice.h: FOLD)
    template <typename ShFn>
    __device__ static __forceinline__ void slice(const Ctx& c, int lane, float x, float y, float z, ShFn f, Out& o) {
        lz_head_slice<false, true, true, true>(c, lane, x, y, z, f, o);
    }
};
template <> struct LzfHead<1> {   // f16: 32-sample slices on v_mfma_f32_32x32x16_f16 (lz_head_f16w_slice.h)
    using Args = LzHead16Args; using Ctx = LzHead16Ctx; using Out = LzHead16wOut; using ShSlot = LzShFromSlot16;
    static constexpr int LDS_WORDS = LZ_HEAD16W_LDS_H8 * 4;
    __device__ static __forceinline__ void stage(const Args& P, float* lds, int, Ctx& c) { lz_head16w_stage(P, reinterpret_cast<lz_h8*>(lds), LZF_WG, c); }
    template <typename ShFn>
    __device__ static __forceinline__ void slice(const Ctx& c, int lane, float x, float y, float z, ShFn f, Out& o) {
        lz_head16w_slice<true, LZ_F16W_PRIO != 0>(c, lane, x, y, z, f, o);
    }
};

// S = samples one ray marches per pass (1, 2, 4, 8 or 16; a slice holds 16 / S rays).  S = 1 is the layout described at the top of the file.
// With few rays a wave would own too few of them to keep the matrix pipe busy for the ~100 dependent passes a ray needs, so the host
// raises S until there are enough 16-sample rows in flight: the loop under the schedule n_step = S (the rows behind a ray's last sample
// in its last pass are the only waste).  Samples of a pass are staged per slot in LDS: the group leader (lane j == 0 of a ray's S
// slots) marches and composites, every slot's lanes evaluate the head.
template <int PREC, int S, int ROWS>
__global__ void __launch_bounds__(LZF_WG, LZF_WG / 256)
lz_k_frame(typename LzfHead<PREC>::Args P, LzFrameK F) {
    using HD = LzfHead<PREC>;
    // f16: always two slot rows (32 slots) per wave -- the 32-sample slice of lz_head16w_slice; f32: one row of 16, several only with S == 1
    static_assert(PREC == 1 ? ROWS == 2 : (ROWS == 1 || S == 1), "slot rows per wave: f16 two; f32 one, or several with one sample per ray and pass");
    constexpr int NS = 16 * ROWS;                                   // ray slots per wave
    using FL = LzfFields<PREC, S, ROWS>;
    constexpr int NF = FL::COUNT;
    constexpr int SF_X = FL::X, SF_Y = FL::Y, SF_Z = FL::Z, SF_TS = FL::TS, SF_OSIG = FL::OSIG, SF_OR = FL::OR, SF_OG = FL::OG, SF_OB = FL::OB,
                  SF_OA0 = FL::OA0, SF_OA1 = FL::OA1, SF_OU = FL::OU, SF_IT = FL::IT, SF_RD = FL::RD;
    (void)SF_RD; (void)SF_X; (void)SF_Y; (void)SF_Z; (void)SF_TS; (void)SF_IT; (void)SF_OSIG; (void)SF_OR; (void)SF_OG; (void)SF_OB; (void)SF_OA0; (void)SF_OA1; (void)SF_OU;
    constexpr int SLOT_WORDS = LZF_WAVES * NF * NS;
    __shared__ __align__(16) float lds[HD::LDS_WORDS + SLOT_WORDS + 4 + LZF_LUT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // S > 1: s = the lane's slot (f32: the four lanes l & 15 of a slot; f16: its two lanes l & 31); S == 1, f32: the slot of its row
    const int s = (S > 1) ? (lane & (NS - 1)) : (lane & 15), q = lane >> 4;
    const bool slot_lane = lane < NS;                               // the lane that marches / composites slot `lane`
    static_assert(alignof(typename HD::Args) <= 8, "lz_k_frame(Args P, LzFrameK F): F sits at round8(sizeof(Args)) of the kernel-argument segment");
    const LzfOut OUT = lzf_out<sizeof(typename HD::Args)>();
    // phase 2 of the reference's cap continues the rays phase 1 parked at max_steps; none parked (or C_eff == max_steps): nothing to stage
    const bool ph2 = F.phase2 != 0;
    if (ph2 && F.state[LZF_P_SIZE] <= 0) return;
    typename HD::Ctx ctx;
    HD::stage(P, lds, q, ctx);
    float* slot = lds + HD::LDS_WORDS + wave * NF * NS;      // this wave's slots: slot[field * 16 + s]
    int* sloti = reinterpret_cast<int*>(slot);
    int* wg_stat = reinterpret_cast<int*>(lds + HD::LDS_WORDS + SLOT_WORDS);   // [0] samples, [1] slices, [2] waves done
    if (lane < NS) sloti[SF_RAY * NS + lane] = -1;
    if (threadIdx.x < 4) wg_stat[threadIdx.x] = 0;
    // bit-spread table of the Morton code (lz_expand_bits) for the march: three LDS reads per probe instead of 24 vector instructions
    uint32_t* mlut = reinterpret_cast<uint32_t*>(lds + HD::LDS_WORDS + SLOT_WORDS + 4);
    const bool use_lut = F.H <= LZF_LUT;
    if (use_lut && threadIdx.x < LZF_LUT) mlut[threadIdx.x] = lz_expand_bits(threadIdx.x);
    __syncthreads();
    const int n_queue = F.state[ph2 ? LZF_P_SIZE : LZF_Q_SIZE];
#define q_head (F.state + (ph2 ? LZF_P_HEAD : LZF_Q_HEAD))        /* (not a variable: it would sit in two scalar registers through every pass) */
    // samples at which a ray still alive is stopped: the per-ray cap ceil(max_steps / S) * S (cap_mode 0), or exactly max_steps in phase 1 and
    // the schedule's C_eff in phase 2 (cap_mode 1)
    const int cap = ph2 ? F.state[LZF_CEFF] : (F.cap_mode ? (int)F.max_steps : (((int)F.max_steps + S - 1) / S) * S);
#define cnt_base (ph2 ? (int)F.max_steps : 0)                     /* samples a ray brings along when it takes a slot */
    // the pointers only the REFILL touches (queue, start / end times, the ray itself): read from the kernel-argument segment at that point like
    // the outputs (LzfOut) -- six more loop-invariant scalar pairs out of the pass loop: the S > 1 kernels drop most of their scalar spills
    // (S = 4: 12 -> 2, S = 16: 73 -> 42), the two-row f16 kernel its one vector spill.  Not with three slot rows, where the same change costs
    // three vector spills instead (kernel_resources.json).  Same-box A/B: f16 8-way tile 0.423 -> 0.418 ms, the full frames within noise.
    constexpr bool KREFILL = ROWS < 3;
#define LZF_RF(name) (KREFILL ? LZF_OUT(OUT, name) : F.name)
    LzMarch m;   // the frame-constant part of LzMarch (init() below sets the per-ray part)
    bool dry = n_queue <= 0;
    int my_samples = 0, my_slices = 0;

    if constexpr (S > 1) {
        const int j = s % S, lead = s - j;              // slot s = step j of the ray whose state sits at slot `lead`
        for (;;) {
            if constexpr (PREC != 1 || LZ_F16W_PRIO != 0) __builtin_amdgcn_s_setprio(2);   // see the S = 1 loop
            // ---------------- refill + march (group leaders): up to S samples per ray into the staging fields ----------------
            const bool leader = slot_lane && j == 0;
            int ray = leader ? sloti[SF_RAY * NS + s] : -1;
            int kk = 0;                                  // samples marched this pass (leaders)
            for (int attempt = 0; attempt < 4; attempt++) {
                const bool need = leader && ray < 0 && !dry;
                const unsigned long long mask = __ballot(need);
                if (mask) {
                    const int first = __ffsll((long long)mask) - 1, take = __popcll(mask);
                    int base = 0;
                    if (lane == first) base = atomicAdd(q_head, take);
                    base = __shfl(base, first, 64);
                    if (need) {
                        const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
                        if (idx < n_queue) {
                            ray = LZF_RF(order)[idx];
                            sloti[SF_RAY * NS + s] = ray;
                            slot[SF_T * NS + s] = LZF_RF(rays_t)[ray];
                            slot[SF_FAR * NS + s] = (LZF_RF(occ) ? LZF_RF(t_end)[ray] : LZF_RF(fars)[ray]);
                            lzf_slot_take(OUT, ph2, ray, slot, sloti, s, NS);
                            lzf_store_sh<PREC, FL::GEO>(LZF_RF(rays_o) + (size_t)ray * 3, LZF_RF(rays_d) + (size_t)ray * 3, slot, s, NS, SF_RD);
                        }
                    }
                    if (base + take >= n_queue) dry = true;
                }
                // ---- march, S candidates of a ray per round on the S lanes of its group (the slot lanes: lane lead + j takes candidate j) ----
                // Every step of the reference's march -- the sample step and the empty-space skip alike -- is t += clamp(t dt_gamma, dt_min,
                // dt_max) (raymarching.cu:907, 919-926): a ray visits a subsequence of ONE fixed chain c_0 = t, c_{i+1} = c_i + step(c_i).  So
                // the group's lanes locate and test the next S chain points TOGETHER (one cell test and one bitfield load deep instead of S),
                // and then all of them replay the serial rule on the S results: an occupied cell is a sample; an empty one at c_i moves the
                // ray to the first chain point at or behind its exit time, exactly what LzMarch::probe's do-while does.  Same cells, same
                // arithmetic per cell (LzMarch::locate / exit_t), same samples.
                {
                    const int gray = __shfl(ray, lead, 64);                      // the group's ray and whether it still has to march this pass
                    const int gkk0 = __shfl(kk, lead, 64);
                    const bool grp = slot_lane && gray >= 0 && gkk0 == 0;
                    if (__ballot(grp)) {
                        float t0 = 0.0f, far = 0.0f;
                        int want = 0, got = 0;
                        if (grp) {
                            if constexpr (FL::GEO) {
                                const float go[3] = {slot[(SF_RD + 3) * NS + lead], slot[(SF_RD + 4) * NS + lead], slot[(SF_RD + 5) * NS + lead]};
                                const float gd[3] = {slot[(SF_RD + 6) * NS + lead], slot[(SF_RD + 7) * NS + lead], slot[(SF_RD + 8) * NS + lead]};
                                m.init(go, gd, slot[SF_RD * NS + lead], slot[(SF_RD + 1) * NS + lead], slot[(SF_RD + 2) * NS + lead], F.bound, F.dt_gamma, F.mf, F.C, F.H, F.grid);
                            } else
                            m.init(F.rays_o + (size_t)gray * 3, F.rays_d + (size_t)gray * 3, slot[SF_RD * NS + lead], slot[(SF_RD + 1) * NS + lead],
                                   slot[(SF_RD + 2) * NS + lead], F.bound, F.dt_gamma, F.mf, F.C, F.H, F.grid);
                            if (use_lut) m.morton_lut = mlut;
                            t0 = slot[SF_T * NS + lead];
                            far = slot[SF_FAR * NS + lead];
                            want = min(S, cap - sloti[SF_CNT * NS + lead]);      // >= 1: a ray at the cap has left its slot
                        }
                        bool more = grp && t0 < far;
                        while (__ballot(more)) {
                            float c = t0;
#pragma unroll
                            for (int i = 0; i < S - 1; i++) c = (i < j) ? c + m.step_at(c) : c;        // chain point j
                            LzMarch::Cell cell;
                            cell.x = cell.y = cell.z = cell.dt = 0.0f;
                            float txo = 0.0f;                                     // < 0: the cell is occupied; else the exit time of the empty cell
                            const bool valid = more && c < far;
                            if (valid) {
                                m.locate(c, cell);
                                txo = m.occupied(cell) ? -1.0f : m.exit_t(c, cell);
                            }
                            const float cn = c + m.step_at(c);                    // chain point j + 1 (= c + cell.dt)
                            // replay of the serial rule over the S candidates, identical on every lane of the group (the chain is recomputed
                            // in step with it -- same operations, same bits -- so only the test results cross lanes: one shuffle per candidate)
                            float skip = -FLT_MAX, t_new = t0, ci = t0;
                            bool stop = false;
                            int rank = -1;
                            if (!__ballot(valid && !(txo < 0.0f))) {
                                // Every candidate inside the box sits in an occupied cell (always, in a dense scene): the replay below would take
                                // the first min(room, candidates in the box) of them in order and stop -- on the room, or on the first chain point
                                // behind `far`, which is the `cn` of the last one taken either way.  One ballot and one shuffle instead of S rounds
                                // (same-box A/B on an 8-way tile: f16 0.420 -> 0.407 ms, f32 1.32 -> 1.30).
                                const unsigned long long vm = __ballot(valid);
                                const int nvalid = __popcll((vm >> lead) & ((1ull << S) - 1ull));     // (slot lanes: lane = slot)
                                const int take = min(want - got, nvalid);
                                const float t_last = __shfl(cn, lead + max(take, 1) - 1, 64);
                                if (more) {
                                    if (j < take) rank = got + j;
                                    got += take;
                                    t_new = t_last;
                                }
                                stop = true;
                            } else {
#pragma unroll      // (not unrolled it spills less -- the f16 variants nothing at all -- and runs 2 % slower on an 8-way tile)
                            for (int i = 0; i < S; i++) {
                                const float cni = ci + m.step_at(ci);
                                const float txi = __shfl(txo, lead + i, 64);
                                const bool live = !stop && ci >= skip;            // not inside an empty cell already stepped over
                                if (live && !(ci < far)) { stop = true; t_new = ci; }               // the ray has left the box
                                else if (live && txi < 0.0f) {
                                    if (i == j) rank = got;
                                    got++;
                                    t_new = cni;
                                    if (got >= want) stop = true;
                                } else if (live) skip = txi;
                                if (!stop && i == S - 1) t_new = cni;
                                ci = cni;
                            }
                            }
                            if (more && !stop) {                                  // first chain point at or behind the last empty cell's exit
                                while (t_new < skip) t_new += m.step_at(t_new);
                            }
                            if (more && rank >= 0) {
                                const int sl = lead + rank;
                                slot[SF_X * NS + sl] = cell.x; slot[SF_Y * NS + sl] = cell.y; slot[SF_Z * NS + sl] = cell.z;
                                slot[SF_DT * NS + sl] = cell.dt;
                                slot[SF_TS * NS + sl] = cn;
                            }
                            t0 = t_new;
                            more = more && got < want && t0 < far;
                        }
                        if (leader && grp) {
                            kk = got;
                            if (kk == 0) {     // the ray left the box without another sample
                                const int c0 = sloti[SF_CNT * NS + s];
                                lzf_ray_end(OUT, ph2, ray, LZF_END_BOX, c0, c0, t0, slot[SF_WS * NS + s], slot[SF_D * NS + s], slot[SF_R * NS + s], slot[SF_G * NS + s],
                                            slot[SF_B * NS + s], slot[SF_A0 * NS + s], slot[SF_A1 * NS + s], slot[SF_U * NS + s]);
                                my_samples += c0 - cnt_base;
                                ray = -1;
                                sloti[SF_RAY * NS + s] = -1;
                            }
                        }
                    }
                }
                if (!__ballot(leader && ray < 0 && !dry)) break;
            }
            if (leader) sloti[SF_IT * NS + s] = kk;      // samples of this pass
            if (!__ballot(kk > 0)) {
                if (dry) break;
                continue;
            }
            // ---------------- head: slot s evaluates sample j of its ray when the leader staged one ----------------
            const int lkk = sloti[SF_IT * NS + lead];
            const bool live = sloti[SF_RAY * NS + lead] >= 0 && j < lkk;
            const float px = live ? slot[SF_X * NS + s] : 0.0f, py = live ? slot[SF_Y * NS + s] : 0.0f, pz = live ? slot[SF_Z * NS + s] : 0.0f;
            typename HD::Out o;
            HD::slice(ctx, lane, px, py, pz, typename HD::ShSlot{slot, lead, NS}, o);
            my_slices += ROWS;
            if constexpr (PREC == 1) {
                // the f16 head leaves a sample's four transcendentals two per lane (LzHead16wOut: rgb[0], rgb[2] on lane half 0; rgb[1], sigma on
                // half 1): every lane parks its own, nothing crosses lanes.  alpha = 1 - exp(-sigma delta) of the sample (raymarching.cu:2197) is
                // computed here by the sigma lane: the leader's serial walk below then costs a handful of instructions per sample
                const int hh = lane >> 5;
                slot[(SF_OR + hh) * NS + s] = o.a;                                  // SF_OR, SF_OG are consecutive
                slot[(hh ? SF_OSIG : SF_OB) * NS + s] = hh ? 1.0f - lz_expf(-o.b * slot[SF_DT * NS + s]) : o.b;
                if (hh == 0) { slot[SF_OA0 * NS + s] = o.ambaud; slot[SF_OA1 * NS + s] = o.eyeatt; slot[SF_OU * NS + s] = o.unc; }
            } else if (q == 0) {
                slot[SF_OSIG * NS + s] = 1.0f - lz_expf(-o.sigma * slot[SF_DT * NS + s]);
                slot[SF_OR * NS + s] = o.rgb[0]; slot[SF_OG * NS + s] = o.rgb[1]; slot[SF_OB * NS + s] = o.rgb[2];
                slot[SF_OA0 * NS + s] = o.ambaud; slot[SF_OA1 * NS + s] = o.eyeatt; slot[SF_OU * NS + s] = o.unc;
            }
            __builtin_amdgcn_wave_barrier();
            // ---------------- composite (lz_k_composite_rays, n_step = S): the leader walks its ray's staged samples ----------------
            if (leader && kk > 0) {
                float ws = slot[SF_WS * NS + s], d = slot[SF_D * NS + s], r = slot[SF_R * NS + s], g = slot[SF_G * NS + s], b = slot[SF_B * NS + s];
                float a0 = slot[SF_A0 * NS + s], a1 = slot[SF_A1 * NS + s], u = slot[SF_U * NS + s], t = slot[SF_T * NS + s];
                int step = 0;
                while (step < kk) {
                    const int sl = s + step;
                    const float alpha = slot[SF_OSIG * NS + sl];
                    const float T = 1 - ws;
                    const float w = alpha * T;
                    ws += w;
                    t = slot[SF_TS * NS + sl];
                    d = lz_fmaf(w, t, d);
                    r = lz_fmaf(w, slot[SF_OR * NS + sl], r);
                    g = lz_fmaf(w, slot[SF_OG * NS + sl], g);
                    b = lz_fmaf(w, slot[SF_OB * NS + sl], b);
                    a0 = a0 + slot[SF_OA0 * NS + sl];
                    a1 = a1 + slot[SF_OA1 * NS + sl];
                    u = lz_fmaf(w, slot[SF_OU * NS + sl], u);
                    if (T < F.T_thresh) break;
                    step++;
                }
                const int c0 = sloti[SF_CNT * NS + s];
                const int cnt = c0 + kk;                                     // marched samples (renderer semantics: the chunk was marched)
                const bool cut = step < kk;                                  // T < T_thresh at sample c0 + step + 1
                const bool short_chunk = kk < min(S, cap - c0);              // the box held fewer samples than the chunk wanted
                if (cut || short_chunk || cnt >= cap) {
                    const int kind = cut ? LZF_END_T : (short_chunk ? LZF_END_BOX : LZF_END_CAP);
                    const int composited = cut ? c0 + step + 1 : cnt;
                    lzf_ray_end(OUT, ph2, ray, kind, composited, cnt, t, ws, d, r, g, b, a0, a1, u);
                    my_samples += (F.cap_mode ? composited : cnt) - cnt_base;
                    sloti[SF_RAY * NS + s] = -1;
                } else {
                    slot[SF_T * NS + s] = t;
                    slot[SF_WS * NS + s] = ws; slot[SF_D * NS + s] = d;
                    slot[SF_R * NS + s] = r; slot[SF_G * NS + s] = g; slot[SF_B * NS + s] = b;
                    slot[SF_A0 * NS + s] = a0; slot[SF_A1 * NS + s] = a1; slot[SF_U * NS + s] = u;
                    sloti[SF_CNT * NS + s] = cnt;
                }
            }
        }
    } else {
        // S = 1.  ROWS slot rows per wave (NS = 16 ROWS slots, slot l lives on lane l): the march, the refill and the compositing run
        // ONCE for all NS slots, the head runs once per row of 16.  ROWS = 2 is for the f16 head, which is bound by vector-instruction
        // issue: the sections that use 16 of 64 lanes are then shared by two slices (outputs of a row wait for the compositing in LDS).
        const int sl = lane;                       // this lane's slot (slot lanes only)
        for (;;) {
            // f32 heads: refill, march and the gather's address work run at a raised wave priority and the slice drops it once its loads are
            // issued (lz_head_gather<YIELD>): a wave that is about to wait on memory gets there first, the matrix phases fill the time
            if constexpr (PREC != 1 || LZ_F16W_PRIO != 0) __builtin_amdgcn_s_setprio(2);
            // ---------------- refill + march: every slot ends with a sample, crossing empty space, or empty with the queue dry ----------------
            int ray = slot_lane ? sloti[SF_RAY * NS + sl] : -1;
            bool have = false;
            float x = 0.0f, y = 0.0f, z = 0.0f;
            for (int attempt = 0; attempt < 4; attempt++) {
                const bool need = slot_lane && ray < 0 && !dry;
                const unsigned long long mask = __ballot(need);
                if (mask) {
                    const int leader = __ffsll((long long)mask) - 1, take = __popcll(mask);
                    int base = 0;
                    if (lane == leader) base = atomicAdd(q_head, take);
                    base = __shfl(base, leader, 64);
                    if (need) {
                        const int idx = base + __popcll(mask & ((1ull << lane) - 1ull));
                        if (idx < n_queue) {
                            ray = LZF_RF(order)[idx];
                            sloti[SF_RAY * NS + sl] = ray;
                            slot[SF_T * NS + sl] = LZF_RF(rays_t)[ray];
                            slot[SF_FAR * NS + sl] = (LZF_RF(occ) ? LZF_RF(t_end)[ray] : LZF_RF(fars)[ray]);
                            lzf_slot_take(OUT, ph2, ray, slot, sloti, sl, NS);
                            lzf_store_sh<PREC, FL::GEO>(LZF_RF(rays_o) + (size_t)ray * 3, LZF_RF(rays_d) + (size_t)ray * 3, slot, sl, NS, SF_RD);
                        }
                    }
                    if (base + take >= n_queue) dry = true;    // wave-uniform
                }
                if (slot_lane && ray >= 0 && !have) {
                    if constexpr (FL::GEO) {
                        const float go[3] = {slot[(SF_RD + 3) * NS + sl], slot[(SF_RD + 4) * NS + sl], slot[(SF_RD + 5) * NS + sl]};
                        const float gd[3] = {slot[(SF_RD + 6) * NS + sl], slot[(SF_RD + 7) * NS + sl], slot[(SF_RD + 8) * NS + sl]};
                        m.init(go, gd, slot[SF_RD * NS + sl], slot[(SF_RD + 1) * NS + sl], slot[(SF_RD + 2) * NS + sl], F.bound, F.dt_gamma, F.mf, F.C, F.H, F.grid);
                    } else
                    m.init(F.rays_o + (size_t)ray * 3, F.rays_d + (size_t)ray * 3, slot[SF_RD * NS + sl], slot[(SF_RD + 1) * NS + sl], slot[(SF_RD + 2) * NS + sl],
                           F.bound, F.dt_gamma, F.mf, F.C, F.H, F.grid);
                    if (use_lut) m.morton_lut = mlut;
                    float t = slot[SF_T * NS + sl], dt = 0.0f;
                    const float far = slot[SF_FAR * NS + sl];
                    // at most LZF_MARCH_PROBES empty cells per attempt: a ray crossing empty space (behind the object, between two blobs) keeps
                    // its slot idle for a few passes instead of stalling the other slots of the wave for the whole crossing
                    int probes = 0;
                    while (t < far && probes < LZF_MARCH_PROBES) {
                        if (m.probe(t, x, y, z, dt)) { have = true; break; }
                        probes++;
                    }
                    if (have) {
                        slot[SF_T * NS + sl] = t;
                        slot[SF_DT * NS + sl] = dt;
                    } else if (t < far) {   // still in empty space: resume from here in the next pass
                        slot[SF_T * NS + sl] = t;
                        x = y = z = 0.0f;
                    } else {        // the ray left the box (renderer.py: the march writes no row, compositing kills the ray on delta == 0)
                        const int c0 = sloti[SF_CNT * NS + sl];
                        lzf_ray_end(OUT, ph2, ray, LZF_END_BOX, c0, c0, t, slot[SF_WS * NS + sl], slot[SF_D * NS + sl], slot[SF_R * NS + sl], slot[SF_G * NS + sl],
                                    slot[SF_B * NS + sl], slot[SF_A0 * NS + sl], slot[SF_A1 * NS + sl], slot[SF_U * NS + sl]);
                        my_samples += c0 - cnt_base;
                        ray = -1;
                        sloti[SF_RAY * NS + sl] = -1;
                        x = y = z = 0.0f;
                    }
                }
                if (!__ballot(slot_lane && ray < 0 && !dry)) break;
            }
            const unsigned long long have_mask = __ballot(have);
            if (!have_mask) {
                if (dry && !__ballot(slot_lane && ray >= 0)) break;     // queue dry and every slot empty: this wave is done
                continue;                                             // slots still crossing empty space (or waiting for a refill)
            }
            // ---------------- head: exactly a slice of the stand-alone head kernel ----------------
            typename HD::Out o;
            if constexpr (PREC == 1) {
                // f16: ONE 32-sample slice over both slot rows (lz_head16w_slice: v_mfma_f32_32x32x16_f16, lane = (slot l & 31, half l >> 5)).
                // A sample's four transcendentals come out two per lane (rgb[0], rgb[2] on half 0; rgb[1], sigma on half 1): every lane parks
                // its own for the compositing below -- OSIG, OR, OG, OB are consecutive fields -- and nothing crosses lanes
                const int rs = lane & 31, hh = lane >> 5;
                // the slot lanes' positions to both lane halves: v_permlane32_swap(a, a) leaves a's low half in both (one VALU instruction where
                // a ds_bpermute costs an address, an LDS round trip and a wait)
                auto lo2 = [](float v) -> float {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                    return __uint_as_float(r[0]);
                };
                const float px = lo2(x), py = lo2(y), pz = lo2(z);
                HD::slice(ctx, lane, px, py, pz, typename HD::ShSlot{slot, rs, NS}, o);
                my_slices += 2;
                slot[(SF_OR + hh) * NS + rs] = o.a;
                slot[(hh ? SF_OSIG : SF_OB) * NS + rs] = o.b;
                if (hh == 0) { slot[SF_OA0 * NS + rs] = o.ambaud; slot[SF_OA1 * NS + rs] = o.eyeatt; slot[SF_OU * NS + rs] = o.unc; }
            } else {
#pragma unroll 1
                for (int row = 0; row < ROWS; row++) {     // one slice per row of 16 slots
                    if (ROWS > 1 && !((have_mask >> (16 * row)) & 0xffffull)) continue;   // no sample in this row
                    const int rs = 16 * row + s;                   // the slot whose sample this lane works on
                    const float px = __shfl(x, rs, 64), py = __shfl(y, rs, 64), pz = __shfl(z, rs, 64);
                    HD::slice(ctx, lane, px, py, pz, typename HD::ShSlot{slot, rs, NS}, o);
                    my_slices++;
                    if (ROWS > 1 && q == 0) {   // park the row's outputs for the compositing below (lanes q == 0 hold valid bits)
                        slot[SF_OSIG * NS + rs] = o.sigma;
                        slot[SF_OR * NS + rs] = o.rgb[0]; slot[SF_OG * NS + rs] = o.rgb[1]; slot[SF_OB * NS + rs] = o.rgb[2];
                        slot[SF_OA0 * NS + rs] = o.ambaud; slot[SF_OA1 * NS + rs] = o.eyeatt; slot[SF_OU * NS + rs] = o.unc;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();     // the parked outputs are read by other lanes of this wave: keep the LDS order
            // ---------------- composite (lz_k_composite_rays, n_step = 1): the slot lanes ----------------
            if (have) {
                float sg, c0, c1, c2, am0, am1, un;
                if constexpr (PREC != 1) { sg = o.sigma; c0 = o.rgb[0]; c1 = o.rgb[1]; c2 = o.rgb[2]; am0 = o.ambaud; am1 = o.eyeatt; un = o.unc; }
                if (ROWS > 1) {
                    sg = slot[SF_OSIG * NS + sl];
                    c0 = slot[SF_OR * NS + sl]; c1 = slot[SF_OG * NS + sl]; c2 = slot[SF_OB * NS + sl];
                    am0 = slot[SF_OA0 * NS + sl]; am1 = slot[SF_OA1 * NS + sl]; un = slot[SF_OU * NS + sl];
                }
                const float dt = slot[SF_DT * NS + sl];
                float ws = slot[SF_WS * NS + sl];
                const float alpha = 1.0f - lz_expf(-sg * dt);
                const float T = 1 - ws;
                const float w = alpha * T;
                ws += w;
                const float t = slot[SF_T * NS + sl] + dt;
                const float d = lz_fmaf(w, t, slot[SF_D * NS + sl]);
                const float r = lz_fmaf(w, c0, slot[SF_R * NS + sl]);
                const float g = lz_fmaf(w, c1, slot[SF_G * NS + sl]);
                const float b = lz_fmaf(w, c2, slot[SF_B * NS + sl]);
                const float a0 = slot[SF_A0 * NS + sl] + am0;
                const float a1 = slot[SF_A1 * NS + sl] + am1;
                const float u = lz_fmaf(w, un, slot[SF_U * NS + sl]);
                const int cnt = sloti[SF_CNT * NS + sl] + 1;
                if (T < F.T_thresh || cnt >= cap) {
                    lzf_ray_end(OUT, ph2, ray, T < F.T_thresh ? LZF_END_T : LZF_END_CAP, cnt, cnt, t, ws, d, r, g, b, a0, a1, u);
                    my_samples += cnt - cnt_base;
                    sloti[SF_RAY * NS + sl] = -1;
                } else {
                    slot[SF_T * NS + sl] = t;
                    slot[SF_WS * NS + sl] = ws; slot[SF_D * NS + sl] = d;
                    slot[SF_R * NS + sl] = r; slot[SF_G * NS + sl] = g; slot[SF_B * NS + sl] = b;
                    slot[SF_A0 * NS + sl] = a0; slot[SF_A1 * NS + sl] = a1; slot[SF_U * NS + sl] = u;
                    sloti[SF_CNT * NS + sl] = cnt;
                }
            }
        }
    }
    // ---------------- statistics: wave -> workgroup (LDS) -> one pair of global atomics by the last wave out ----------------
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_samples += __shfl_xor(my_samples, off, 64);
    if (lane == 0) {
        atomicAdd(&wg_stat[0], my_samples);
        atomicAdd(&wg_stat[1], my_slices);
        __threadfence_block();
        if (atomicAdd(&wg_stat[2], 1) == LZF_WAVES - 1) {
            const int a = atomicAdd(&wg_stat[0], 0), b = atomicAdd(&wg_stat[1], 0);
            if (a) atomicAdd(F.state + LZF_SAMPLES, a);
            if (b) atomicAdd(F.state + LZF_ROWS, 16 * b);
        }
    }
}

#undef q_head
#undef cnt_base
#undef LZF_RF

// ---- the reference's cap (cap_mode 1): histogram of L, schedule replay, marched counts ------------------------------------------------
// cap_ws: [0 .. max_steps] histogram of L (bin max_steps = rays alive at the cap); behind it the schedule tables, see lzf_ws_*
__host__ __device__ __forceinline__ int lzf_ws_chunk_end(int max_steps) { return max_steps + 1; }     // [max_steps + 9]: index c = 1-based sample

// The reference's loop on the counts alone (renderer.py:503-548): n_alive -> n_step = max(min(N // n_alive, 8), 1) -> step += n_step while
// step < max_steps, with n_alive at boundary B = rays with L >= B (suffix sums of the histogram).  n_alive never grows along B, so n_step
// never shrinks: the boundaries the loop visits fall into at most 8 runs of constant stride k = 1 .. 8, each ending where n_step first
// exceeds k (found data-parallel), and one lane walks the runs in closed form instead of chasing ~max_steps dependent LDS reads.
// Writes C_eff (the boundary the loop stops at), the iteration count, chunk_end[c] = the boundary closing the chunk of sample c (1-based),
// and empties phase 2's queue when C_eff <= max_steps (the parked rays are complete as they are).  lds: LZF_SCHED_LDS_INTS.
#define LZF_SCHED_LDS_INTS(max_steps) ((max_steps) + 9 + 64)
__device__ __forceinline__ int lzf_n_step(uint32_t N, int n_alive) {   // max(min(N // n_alive, 8), 1) without the division
    const unsigned long long na = (unsigned long long)n_alive;
    int k = 1;
#pragma unroll
    for (int j = 2; j <= 8; j++) k += (na * j <= (unsigned long long)N) ? 1 : 0;
    return k;
}
__device__ void lzf_schedule(const LzFrameK& F, int* lds) {
    const int ms = (int)F.max_steps, n = ms + 9, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = (nt + 63) >> 6;
    const uint32_t N = F.N_total ? F.N_total : F.N;
    int* alive = lds;                    // [n]
    int* wtot = lds + n;                 // [16] wave totals of the scan
    int* first = lds + n + 16;           // [10]: first[k] = first boundary whose n_step exceeds k (k = 1 .. 8; 0 unused); first[9] = first boundary nobody is alive at
    int* run = lds + n + 32;             // [8][3] runs of the walk: start, stride, steps; run[24] = count, [25] = C_eff, [26] = iterations
    for (int i = tid; i < n; i += nt) alive[i] = i <= ms ? __hip_atomic_load(F.cap_ws + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    if (tid < 10) first[tid] = n;
    if (tid < 16) wtot[tid] = 0;
    __syncthreads();
    // suffix sums: a contiguous stretch per lane, lanes of a wave by shuffles, waves through LDS
    const int E = (n + nt - 1) / nt, lo = min(tid * E, n), hi = min(lo + E, n);
    int tot = 0;
    for (int i = hi - 1; i >= lo; i--) { tot += alive[i]; alive[i] = tot; }
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_down(incl, off, 64);
        if (lane + off < 64) incl += v;
    }
    if (lane == 0) wtot[wave] = incl;
    __syncthreads();
    int add = incl - tot;
    for (int w = wave + 1; w < nw; w++) add += wtot[w];
    for (int i = lo; i < hi; i++) alive[i] += add;
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        const int na = alive[i];
        if (na <= 0) { atomicMin(&first[9], i); continue; }
        const int k = lzf_n_step(N, na);
        const int kp = i > 0 ? (alive[i - 1] > 0 ? lzf_n_step(N, alive[i - 1]) : 9) : 0;     // n_step just below: only the lanes where it changes write
        for (int j = kp; j < k; j++) if (j >= 1) atomicMin(&first[j], i);
    }
    __syncthreads();
    if (tid == 0) {
        const int stop = min(ms, first[9]);              // the loop ends at max_steps, or where n_alive == 0
        int B = 0, K = 0, r = 0;
        while (B < stop && r < 8) {
            const int k = lzf_n_step(N, alive[B]);
            const int lim = min(stop, first[k]);         // boundaries below `lim` step by k
            const int m = (lim - B + k - 1) / k;         // >= 1: n_step(B) == k means B < first[k]
            run[3 * r] = B; run[3 * r + 1] = k; run[3 * r + 2] = m;
            B += m * k;
            K += m;
            r++;
        }
        run[24] = r; run[25] = B; run[26] = K;
        F.state[LZF_CEFF] = B;
        F.state[LZF_SCHED_K] = K;
        if (B <= ms) atomicExch(F.state + LZF_P_SIZE, 0);
    }
    __syncthreads();
    int* chunk_end = F.cap_ws + lzf_ws_chunk_end(ms);
    const int nr = run[24], c_eff = run[25];
    for (int c = 1 + tid; c < n; c += nt) {
        int e = c;                                       // behind the loop's end: unused
        if (c <= c_eff) {
            for (int r = 0; r < nr; r++) {
                const int b0 = run[3 * r], k = run[3 * r + 1], b1 = b0 + k * run[3 * r + 2];
                if (c > b0 && c <= b1) e = b0 + (c - b0 + k - 1) / k * k;
            }
        }
        chunk_end[c] = e;
    }
}
__global__ void __launch_bounds__(256) lz_k_frame_schedule(LzFrameK F) {
    extern __shared__ int sc_lds[];
    lzf_schedule(F, sc_lds);
}

// one lane per ray: LDS histogram of ray_last per workgroup, flushed with one atomic per non-empty bin; the rays parked at the cap are
// compacted into the queue (order[]; phase 1 has consumed it) with one global atomic per workgroup.  SCHED (a whole frame, nothing to
// exchange): the last workgroup to flush replays the schedule right here (one launch less).
template <bool SCHED>
__global__ void __launch_bounds__(1024) lz_k_frame_cap_hist(LzFrameK F) {
    extern __shared__ int ch_lds[];                      // [max_steps + 1] bins, then 4 words; SCHED: lzf_schedule's LZF_SCHED_LDS_INTS
    // the ticket lives OUTSIDE the dynamic array: lzf_schedule's first statement rewrites ch_lds[0 .. max_steps + 8], and a wave late to
    // read a ticket kept in there would see the 0 a faster wave had already stored and leave (its stretch of the scan skipped)
    __shared__ int ticket;
    const int ms = (int)F.max_steps, bins = ms + 1;
    int* cnt = ch_lds + bins;                            // [0] parked rays of this workgroup, [1] their base in the queue, [2] cursor
    for (int i = threadIdx.x; i < bins + 4; i += blockDim.x) ch_lds[i] = 0;
    __syncthreads();
    // few, fat workgroups (grid-stride): every workgroup ends with one global atomic per non-empty bin, and same-address atomics serialise
    // in the L2 (one workgroup per 256 rays: 1024 x ~100 bins took 80 us)
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t n0 = blockIdx.x * blockDim.x; n0 < F.N; n0 += 8 * stride) {      // (workgroup-uniform trip count)
        int Lv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {                    // eight loads in flight (the words come from HBM: the kernel before wrote them on other XCDs)
            const uint32_t n = n0 + u * stride + threadIdx.x;
            Lv[u] = n < F.N ? F.ray_last[n] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t n = n0 + u * stride + threadIdx.x;
            int L = Lv[u];
            if (n < F.N) L = L < 0 ? 0 : (L > ms ? ms : L);
            // neighbouring rays end at the same boundary more often than not: one LDS atomic per distinct value of the wave, not per lane
            unsigned long long todo = __ballot(L >= 0);
            while (todo) {
                const int v = __shfl(L, __ffsll((long long)todo) - 1, 64);
                const unsigned long long same = __ballot(L == v);
                if ((threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(v == ms ? &cnt[0] : &ch_lds[v], __popcll(same));
                todo &= ~same;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && cnt[0] > 0) {
        cnt[1] = atomicAdd(F.state + LZF_P_SIZE, cnt[0]);
        atomicAdd(F.cap_ws + ms, cnt[0]);
    }
    __syncthreads();
    if (cnt[0] > 0) {                                    // second sweep (the words are cache-hot): queue positions for the parked rays
        for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < F.N; n += stride)
            if (F.ray_last[n] >= ms) F.order[cnt[1] + atomicAdd(&cnt[2], 1)] = (int)n;
    }
    for (int i = threadIdx.x; i < ms; i += blockDim.x) {
        const int h = ch_lds[i];
        if (h) atomicAdd(F.cap_ws + i, h);
    }
    if constexpr (SCHED) {
        // last workgroup out replays the schedule.  Everything it reads was written with agent-scope atomics (performed at the
        // coherence point, past the per-XCD L2s) and is read back with agent-scope atomic loads, so ordering is all that is needed: every
        // lane's atomics acknowledged (vmcnt 0) before the ticket is taken.  NOT __threadfence(): an agent-scope fence on this part writes
        // back and invalidates the XCD's whole L2 -- with the frame's outputs dirty in it that cost ~30 us per workgroup that did it.
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) ticket = atomicAdd(F.state + LZF_TICKET, 1);
        __syncthreads();
        if (ticket != (int)gridDim.x - 1) return;        // workgroup-uniform
        lzf_schedule(F, ch_lds);
    }
}

// one lane per ray, only with ray_counts: a ray the compositing cut at sample tau (count flagged negative, t behind that sample in rays_t)
// was marched by the reference to the end of the chunk holding tau: count = min(box samples, chunk_end[tau])
__global__ void __launch_bounds__(256) lz_k_frame_counts(LzFrameK F) {
    __shared__ uint32_t mlut[LZF_LUT];
    __shared__ int extra_sum;
    mlut[threadIdx.x] = lz_expand_bits(threadIdx.x);
    if (threadIdx.x == 0) extra_sum = 0;
    __syncthreads();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    int extra = 0;
    if (n < F.N) {
        const int c = F.ray_counts[n];
        if (c < 0) {
            const int tau = -c;
            const int end = F.cap_ws[lzf_ws_chunk_end((int)F.max_steps) + tau];
            LzMarch m;
            m.init(F.rays_o + (size_t)n * 3, F.rays_d + (size_t)n * 3, F.bound, F.dt_gamma, F.max_steps, F.C, F.H, F.grid);
            if (F.H <= LZF_LUT) m.morton_lut = mlut;
            float t = F.rays_t[n], x, y, z, dt;
            const float far = F.occ ? F.t_end[n] : F.fars[n];
            while (t < far && tau + extra < end) {
                if (m.probe(t, x, y, z, dt)) { t += dt; extra++; }
            }
            F.ray_counts[n] = tau + extra;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) extra += __shfl_xor(extra, off, 64);
    if ((threadIdx.x & 63) == 0 && extra) atomicAdd(&extra_sum, extra);
    __syncthreads();
    if (threadIdx.x == 0 && extra_sum) atomicAdd(F.state + LZF_SAMPLES, extra_sum);
}

// ---- bounds of the occupied cells (once per bitfield; lz_frame_fused.occupied_aabb) ---------------------------------------------------
// ws [C][6] int32, preset to -1: per cascade level max(H - 1 - x), max(H - 1 - y), max(H - 1 - z), max(x), max(y), max(z) over the set
// bits (bit index = level * H^3 + morton(x, y, z), raymarching.cu:267-300, 890-895)
__global__ void __launch_bounds__(256) lz_k_occupied_cells(const uint8_t* __restrict__ bits, uint32_t C, uint32_t H, int* __restrict__ ws) {
    __shared__ int lm[8 * 6];
    if (threadIdx.x < 48) lm[threadIdx.x] = -1;
    __syncthreads();
    const uint32_t H3 = H * H * H;
    const uint64_t n_bytes = ((uint64_t)C * H3 + 7) / 8, n_chunks = (n_bytes + 15) / 16;
    // a thread takes 16 consecutive bytes at a time and keeps the extremes of the level it is in in registers: six LDS atomics per chunk
    // and level, not per set bit (a dense grid has 2 M of them)
    for (uint64_t ch = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < n_chunks; ch += (uint64_t)gridDim.x * blockDim.x) {
        int cur = -1, m[6] = {-1, -1, -1, -1, -1, -1};
        auto flush = [&]() {
            if (cur >= 0 && m[3] >= 0) {
#pragma unroll
                for (int a = 0; a < 6; a++) atomicMax(lm + cur * 6 + a, m[a]);
            }
        };
        for (uint32_t j = 0; j < 16; j++) {
            const uint64_t b = ch * 16 + j;
            if (b >= n_bytes) break;
            uint32_t v = bits[b];
            while (v) {
                const int k = __ffs((int)v) - 1;
                v &= v - 1u;
                const uint64_t index = b * 8 + (uint32_t)k;
                const int level = (int)(index / H3);
                if (level >= (int)C) break;
                if (level != cur) {
                    flush();
                    cur = level;
#pragma unroll
                    for (int a = 0; a < 6; a++) m[a] = -1;
                }
                const uint32_t mort = (uint32_t)(index % H3);
                const int x = (int)lz_morton3_inv(mort), y = (int)lz_morton3_inv(mort >> 1), z = (int)lz_morton3_inv(mort >> 2);
                m[0] = max(m[0], (int)H - 1 - x); m[1] = max(m[1], (int)H - 1 - y); m[2] = max(m[2], (int)H - 1 - z);
                m[3] = max(m[3], x); m[4] = max(m[4], y); m[5] = max(m[5], z);
            }
        }
        flush();
    }
    __syncthreads();
    if (threadIdx.x < 6 * C && lm[threadIdx.x] >= 0) atomicMax(ws + threadIdx.x, lm[threadIdx.x]);
}
// world-space box: cell n of level l spans ((n .. n + 1) / H * 2 - 1) * min(2^l, bound) (raymarching.cu:409-417); `margin` cells of the
// level's own size (and never less than four of the march's longest steps) on every side.  A side that reaches the rim of the outermost level is open (positions are clamped to the bound
// before the cell test, so the rim cells also stand for everything outside).  No occupied cell: a box at +FLT_MAX that no ray reaches.
__global__ void lz_k_occupied_box(const int* __restrict__ ws, uint32_t C, uint32_t H, float bound, int margin, float* __restrict__ box) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    bool any = false;
    // never less than four of the march's longest steps (dt_max = 2 sqrt(3) 2^(C - 1) / H, raymarching.cu:866): the march reaches the
    // first occupied cell through at least that many ordinary cell tests
    const float steps4 = 4.0f * (2 * LZ_SQRT3F * (float)(1u << (C - 1)) / (float)H);
    for (uint32_t l = 0; l < C; l++) {
        const float mb = lz_fminf(lz_scalbnf(1.0f, (int)l), bound);
        const float pad = lz_fmaxf((float)margin * (2.0f * mb / (float)H), steps4);
        for (int a = 0; a < 3; a++) {
            const int mn_r = ws[l * 6 + a], mx = ws[l * 6 + 3 + a];
            if (mx < 0) continue;
            any = true;
            const int c0 = (int)H - 1 - mn_r, c1 = mx + 1;
            float w0 = ((float)c0 / (float)H * 2.0f - 1.0f) * mb - pad, w1 = ((float)c1 / (float)H * 2.0f - 1.0f) * mb + pad;
            if (l == C - 1 && w0 <= -mb) w0 = -FLT_MAX;
            if (l == C - 1 && w1 >= mb) w1 = FLT_MAX;
            lo[a] = lz_fminf(lo[a], w0);
            hi[a] = lz_fmaxf(hi[a], w1);
        }
    }
    for (int a = 0; a < 3; a++) {
        box[a] = any ? lo[a] : FLT_MAX;
        box[3 + a] = any ? hi[a] : FLT_MAX;
    }
}

extern "C" int lz_occupied_bounds(const uint8_t* bitfield, uint32_t C, uint32_t H, float bound, uint32_t margin, int32_t* workspace, float* aabb6,
                                  lz_stream_t stream) {
    LZ_REQUIRE(bitfield && workspace && aabb6, LZ_ERR_BAD_ARGUMENT, "occupied_bounds: null tensor");
    LZ_REQUIRE(C >= 1 && C <= 8 && H >= 1 && H <= 1024 && bound > 0.0f && margin <= 64, LZ_ERR_BAD_ARGUMENT,
               "occupied_bounds: cascade in [1, 8], grid size in [1, 1024], bound > 0, margin <= 64");
    hipStream_t st = lz_st(stream);
    hipError_t rc = hipMemsetAsync(workspace, 0xff, 48 * sizeof(int32_t), st);
    if (rc != hipSuccess) { lz_set_error("occupied_bounds: memset: %s", hipGetErrorString(rc)); return (int)rc; }
    const uint64_t n_bytes = ((uint64_t)C * H * H * H + 7) / 8;
    uint32_t nb = (uint32_t)lz_div_up(n_bytes, (uint64_t)256 * 16);   // one 16-byte chunk per thread
    nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    hipLaunchKernelGGL(lz_k_occupied_cells, dim3(nb), dim3(256), 0, st, bitfield, C, H, workspace);
    hipLaunchKernelGGL(lz_k_occupied_box, dim3(1), dim3(64), 0, st, workspace, C, H, bound, (int)margin, aabb6);
    LZ_CHECK_LAUNCH("occupied_bounds");
    return LZ_OK;
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
static void lzf_level_tables(const lz_head_params* p, float* scale, uint32_t* res) {
    for (int l = 0; l < 12; l++) {  // gridencoder.cu:125-126 on the host, same libm call as the CPU checker
        const float sc = exp2f((float)l * p->S) * (float)p->H - 1.0f;
        scale[l] = sc;
        res[l] = (uint32_t)ceilf(sc) + 1u;
    }
}

static int lzf_fill(const lz_frame_fused* f, LzFrameK& K) {
    K.rays_o = f->rays_o; K.rays_d = f->rays_d; K.grid = f->grid; K.aabb = f->aabb;
    K.nears = f->nears; K.fars = f->fars; K.rays_t = f->rays_t;
    K.order = f->order; K.state = f->state; K.keys = f->keys;
    K.weights_sum = f->weights_sum; K.depth = f->depth; K.image = f->image; K.amb0_sum = f->amb_aud_sum; K.amb1_sum = f->amb_eye_sum;
    K.unc_sum = f->unc_sum; K.out = f->out; K.bg = f->bg; K.out_rgb24 = f->out_rgb24; K.ray_counts = f->ray_counts;
    K.bg_scalar = f->bg_scalar; K.bound = f->bound; K.dt_gamma = f->dt_gamma; K.T_thresh = f->T_thresh; K.min_near = f->min_near;
    K.N = f->N; K.max_steps = f->max_steps; K.C = f->C; K.H = f->H;
    K.noises = f->noises;
    K.occ = f->occupied_aabb; K.t_end = f->t_end;
    K.ray_last = f->ray_last; K.cap_ws = f->cap_ws; K.cap_mode = f->cap_mode; K.phase2 = 0; K.N_total = f->N_total;
    K.mf = lz_march_frame(f->bound, f->max_steps, f->C, f->H);
    return LZ_OK;
}

static int lzf_check(const lz_frame_fused* f, const char* who) {
    const lz_head_params* p = &f->head;
    LZ_REQUIRE(f->rays_o && f->rays_d && f->grid && f->aabb && f->nears && f->fars && f->rays_t && f->order && f->state && f->keys &&
                   f->weights_sum && f->depth && f->image && f->amb_aud_sum && f->amb_eye_sum && f->unc_sum && f->out,
               LZ_ERR_BAD_ARGUMENT, "%s: incomplete lz_frame_fused", who);
    LZ_REQUIRE(p->emb_xy && p->emb_yz && p->emb_xz && p->offsets && p->packed && p->enc_a, LZ_ERR_BAD_ARGUMENT, "%s: incomplete lz_head_params", who);
    LZ_REQUIRE(p->testing, LZ_ERR_UNSUPPORTED, "%s: inference only (head.testing must be 1)", who);
    LZ_REQUIRE(p->precision >= 0 && p->precision <= 2, LZ_ERR_BAD_ARGUMENT, "%s: precision must be 0 (f32), 1 (f16) or 2 (f32, folded geo)", who);
    LZ_REQUIRE(f->C >= 1 && f->C <= 8 && f->H > 0, LZ_ERR_BAD_ARGUMENT, "%s: cascade must be in [1, 8]", who);
    LZ_REQUIRE(!f->occupied_aabb || f->t_end, LZ_ERR_BAD_ARGUMENT, "%s: occupied_aabb needs the t_end scratch buffer", who);
    LZ_REQUIRE(f->cap_mode == LZ_FRAME_CAP_PER_RAY || f->cap_mode == LZ_FRAME_CAP_REFERENCE, LZ_ERR_BAD_ARGUMENT, "%s: cap_mode must be 0 (per ray) or 1 (the reference's schedule)", who);
    if (f->cap_mode == LZ_FRAME_CAP_REFERENCE) {
        LZ_REQUIRE(f->ray_last && f->cap_ws, LZ_ERR_BAD_ARGUMENT, "%s: cap_mode 1 needs the ray_last and cap_ws buffers", who);
        LZ_REQUIRE(f->max_steps <= LZF_CAP_MAX_STEPS, LZ_ERR_UNSUPPORTED, "%s: cap_mode 1 supports max_steps <= %d", who, LZF_CAP_MAX_STEPS);
        LZ_REQUIRE(f->N_total == 0 || f->N_total >= f->N, LZ_ERR_BAD_ARGUMENT, "%s: N_total is the ray count of the whole frame (>= N)", who);
    }
    // the heads' gathers run without range clamps here (lz_head_gather<IN_RANGE>): every sample the march emits is clamped to ITS bound
    LZ_REQUIRE(f->bound > 0.0f && f->bound <= p->bound, LZ_ERR_BAD_ARGUMENT,
               "%s: the march's bound must not exceed the head's (the reference uses one `bound` for both, renderer.py:94, network.py:100)", who);
    return LZ_OK;
}

// the persistent kernel over the queue (phase 1) or over the rays parked at the cap (K.phase2)
static int lzf_launch_persistent(const lz_frame_fused* f, const LzFrameK& K, hipStream_t st) {
    const lz_head_params* p = &f->head;
    const int n_cu = lz_cu_count();   // of the current device, per call (cached per device)
    // one workgroup per CU (the weights fill most of its LDS); fewer when there are not enough rays for one slot row per wave
    uint32_t grid = lz_div_up(f->N, 16 * 4);   // at least 4 waves' worth of slots per workgroup
    if (grid > (uint32_t)n_cu) grid = (uint32_t)n_cu;
    // samples per ray and pass: 1 when there are enough rays to give every wave of the chip 16 of them twice over; doubled until the
    // 16-sample rows in flight (N * S / 16) reach that number otherwise (f->steps_per_pass overrides: 1, 2, 4, 8 or 16)
    uint32_t S = f->steps_per_pass;
    if (S == 0) {
        // (round 4, with the batched march: rank 0's tile of a 512^2 frame sharded 8 / 4 ways, f16 kernel ms at S = 1 / 2 / 4: 0.573 / 0.454 /
        // 0.421 and 0.733 / 0.705 / 0.768; f32: 2.22 / 1.57 / 1.30 and 2.97 / 2.57 / 2.53 -- both heads want the rows in flight twice over)
        // f16 (32 slots per wave): the slots filled 1.5 times over -- with exactly one ray per slot nothing is ever refilled and the frame
        // ends on emptying slices -- for S <= 2, once over from S = 4 on (round 5, tools/spp_sweep.sh, rank 0's interleaved tile of a 512^2
        // frame, ms at S = 1 / 2 / 4 / 8 / 16: 2-way 1.12 / 1.07 / 1.12 / 1.29 / 1.91, 4-way 0.80 / 0.67 / 0.63 / 0.72 / 1.05, 8-way 0.64 / 0.49 /
        // 0.39 / 0.41 / 0.58, 16-way 0.60 / 0.41 / 0.32 / 0.28 / 0.37: the rule picks the best S at every size)
        const uint64_t slots = (uint64_t)n_cu * LZF_WAVES * 32;
        const uint64_t want = p->precision == 1 ? slots * 3 / 2 : (uint64_t)n_cu * LZF_WAVES * 16 * 2;
        S = 1;
        while (S < 16 && (uint64_t)f->N * S < want) S *= 2;
        if (p->precision == 1 && S >= 8 && (uint64_t)f->N * (S / 2) >= slots) S /= 2;
    }
    LZ_REQUIRE(S == 1 || S == 2 || S == 4 || S == 8 || S == 16, LZ_ERR_BAD_ARGUMENT, "frame_render: steps_per_pass must be 0 (auto), 1, 2, 4, 8 or 16");
#define LZF_LAUNCH(PREC, SS) hipLaunchKernelGGL((lz_k_frame<PREC, SS, 1>), dim3(grid), dim3(LZF_WG), 0, st, a, K)
#define LZF_SWITCH(PREC)                                                        \
    switch (S) {                                                                \
        case 1: LZF_LAUNCH(PREC, 1); break;                                     \
        case 2: LZF_LAUNCH(PREC, 2); break;                                     \
        case 4: LZF_LAUNCH(PREC, 4); break;                                     \
        case 8: LZF_LAUNCH(PREC, 8); break;                                     \
        default: LZF_LAUNCH(PREC, 16); break;                                   \
    }
    if (p->precision == 1) {
        LzHead16Args a;
        a.emb[0] = p->emb_xy; a.emb[1] = p->emb_yz; a.emb[2] = p->emb_xz;
        a.offsets = p->offsets; a.packed = reinterpret_cast<const lz_h8*>(p->packed); a.enc_a = p->enc_a; a.ind_code = p->ind_code;
        a.eye = p->eye; a.bound = p->bound;
        lzf_level_tables(p, a.scale, a.res);
        // f16: two slot rows (32 slots) per wave at every S -- the head works on 32-sample slices (lz_head16w_slice), and the march / refill /
        // compositing sections cost the same instructions for 16 or 32 active lanes.  (Rounds 2-4 also had one- and three-row layouts on
        // 16-sample slices; the 32x32x16 slice halves the MFMA issue slots instead: DESIGN 4.1b.)
        switch (S) {
            case 1: hipLaunchKernelGGL((lz_k_frame<1, 1, 2>), dim3(grid), dim3(LZF_WG), 0, st, a, K); break;
            case 2: hipLaunchKernelGGL((lz_k_frame<1, 2, 2>), dim3(grid), dim3(LZF_WG), 0, st, a, K); break;
            case 4: hipLaunchKernelGGL((lz_k_frame<1, 4, 2>), dim3(grid), dim3(LZF_WG), 0, st, a, K); break;
            case 8: hipLaunchKernelGGL((lz_k_frame<1, 8, 2>), dim3(grid), dim3(LZF_WG), 0, st, a, K); break;
            default: hipLaunchKernelGGL((lz_k_frame<1, 16, 2>), dim3(grid), dim3(LZF_WG), 0, st, a, K); break;
        }
    } else {
        LzHeadArgs a;
        a.emb[0] = p->emb_xy; a.emb[1] = p->emb_yz; a.emb[2] = p->emb_xz;
        a.offsets = p->offsets; a.packed = reinterpret_cast<const float*>(p->packed); a.enc_a = p->enc_a; a.ind_code = p->ind_code; a.eye = p->eye;
        a.bound = p->bound; a.testing = 1;
        lzf_level_tables(p, a.scale, a.res);
        if (p->precision == 2) { LZF_SWITCH(2) } else { LZF_SWITCH(0) }
    }
#undef LZF_SWITCH
#undef LZF_LAUNCH
    return LZ_OK;
}

extern "C" int lz_frame_finish(const lz_frame_fused* f, lz_stream_t stream) {
    LZ_REQUIRE(f, LZ_ERR_BAD_ARGUMENT, "frame_finish: null");
    LZ_REQUIRE(f->cap_mode == LZ_FRAME_CAP_REFERENCE, LZ_ERR_BAD_ARGUMENT, "frame_finish: only after lz_frame_render with cap_mode 1");
    if (f->N == 0 && f->state) return LZ_OK;       // an empty tile (see lz_frame_render)
    int rc = lzf_check(f, "frame_finish");
    if (rc != LZ_OK) return rc;
    hipStream_t st = lz_st(stream);
    LzFrameK K;
    lzf_fill(f, K);
    if (f->defer_finish) hipLaunchKernelGGL(lz_k_frame_schedule, dim3(1), dim3(256), LZF_SCHED_LDS_INTS(f->max_steps) * sizeof(int), st, K);
    K.phase2 = 1;
    rc = lzf_launch_persistent(f, K, st);
    if (rc != LZ_OK) return rc;
    if (f->ray_counts) hipLaunchKernelGGL(lz_k_frame_counts, dim3(lz_div_up(f->N, 256)), dim3(256), 0, st, K);
    LZ_CHECK_LAUNCH("frame_finish");
    return LZ_OK;
}

extern "C" int lz_frame_render(const lz_frame_fused* f, lz_timing* timing, lz_stream_t stream) {
    LZ_REQUIRE(f, LZ_ERR_BAD_ARGUMENT, "frame_render: null");
    hipStream_t st = lz_st(stream);
    if (f->N == 0 && f->state) {
        // A rank whose tile of the frame holds no ray (a frame with fewer row blocks than ranks): nothing to render -- the per-ray
        // buffers may be null -- but its words of the histogram exchange must read zero, and the state must say "done, no samples".
        hipError_t e = hipMemsetAsync(f->state, 0, LZ_FRAME_STATE_INTS * sizeof(int32_t), st);
        if (e == hipSuccess && f->cap_mode == LZ_FRAME_CAP_REFERENCE && f->cap_ws) {
            LZ_REQUIRE(f->max_steps <= LZF_CAP_MAX_STEPS, LZ_ERR_UNSUPPORTED, "frame_render: cap_mode 1 supports max_steps <= %d", LZF_CAP_MAX_STEPS);
            e = hipMemsetAsync(f->cap_ws, 0, LZ_FRAME_CAP_WS_INTS((size_t)f->max_steps) * sizeof(int32_t), st);
        }
        if (e != hipSuccess) { lz_set_error("frame_render: memset: %s", hipGetErrorString(e)); return (int)e; }
        return LZ_OK;
    }
    int rc = lzf_check(f, "frame_render");
    if (rc != LZ_OK) return rc;
    LzFrameK K;
    lzf_fill(f, K);
    hipError_t hrc = hipMemsetAsync(f->state, 0, LZ_FRAME_STATE_INTS * sizeof(int32_t), st);
    if (hrc != hipSuccess) { lz_set_error("frame_render: memset: %s", hipGetErrorString(hrc)); return (int)hrc; }
    const bool ref_cap = f->cap_mode == LZ_FRAME_CAP_REFERENCE;      // (cap_ws is zeroed by lz_k_frame_prepare)
    const uint32_t nb = lz_div_up(f->N, LZF_PREP_WG);
    hipLaunchKernelGGL(lz_k_frame_prepare, dim3(nb), dim3(LZF_PREP_WG), 0, st, K);
    hipLaunchKernelGGL(lz_k_frame_scatter, dim3(nb), dim3(LZF_PREP_WG), 0, st, K);
    if (timing) (void)lz_timing_mark(timing, 0, stream);    // the event pair brackets the persistent kernel alone
    rc = lzf_launch_persistent(f, K, st);
    if (rc != LZ_OK) return rc;
    if (timing) (void)lz_timing_mark(timing, 1, stream);
    if (ref_cap) {
        const size_t lds = LZF_SCHED_LDS_INTS((size_t)f->max_steps) * sizeof(int);      // >= the histogram's max_steps + 5
        uint32_t hb = lz_div_up(f->N, 1024 * 8);            // 8 rays per lane, at most 32 workgroups
        hb = hb < 1 ? 1 : (hb > 32 ? 32 : hb);
        if (f->defer_finish) hipLaunchKernelGGL(lz_k_frame_cap_hist<false>, dim3(hb), dim3(1024), lds, st, K);
        else hipLaunchKernelGGL(lz_k_frame_cap_hist<true>, dim3(hb), dim3(1024), lds, st, K);
    }
    LZ_CHECK_LAUNCH("frame_render");
    if (ref_cap && !f->defer_finish) return lz_frame_finish(f, stream);
    return LZ_OK;
}
