// lz_render.hip -- native host-side driver of the device-resident inference loop.
//
// The reference drives its loop from Python (nerf_triplane/renderer.py:503-548): per iteration ~20 kernel launches
// and one device->host synchronisation.  Here one C call enqueues `n_iterations` iterations of
//     lz_loop_march (state advance + compaction + march) -> lz_triplane_head_forward -> lz_loop_composite (+ state commit)
// back to back on the caller's stream (3 launches each, no host round trip, no Python between launches).  Iterations
// enqueued after the frame finished are no-ops on the device (every kernel is bounded by the device-side state).
//
// Optional timing: an lz_timing object owns pairs of HIP events; when one is passed, every head launch is bracketed
// by an event pair ON THE LAUNCH STREAM, so a benchmark gets the dominant kernel's per-launch duration from the very
// launches it times end to end (no separate pass, no profiler attached).
#include <vector>

#include "lz_common.h"

struct lz_timing {
    std::vector<hipEvent_t> ev;  // 2 per pair
    uint32_t used = 0;           // pairs recorded
};

extern "C" int lz_timing_create(uint32_t n_pairs, lz_timing** out) {
    LZ_REQUIRE(out, LZ_ERR_BAD_ARGUMENT, "timing_create: null out");
    lz_timing* t = new lz_timing();
    t->ev.resize((size_t)n_pairs * 2);
    for (auto& e : t->ev) {
        // timing only: no system-scope cache flush at the record (the default event releases to the host, i.e. an L2 write-back
        // in front of and behind every head launch: 0.3 ms of an 11.7 ms frame)
        hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
        if (rc != hipSuccess) {
            lz_set_error("timing_create: hipEventCreate: %s", hipGetErrorString(rc));
            delete t;
            return (int)rc;
        }
    }
    *out = t;
    return LZ_OK;
}

// one wave: lane r polls flag r (system-scope loads: the writers are other devices / processes) with a sleep between polls; every lane
// leaves after at most max_polls polls, so the kernel always ends
__global__ void __launch_bounds__(64) lz_k_wait_flags(const int32_t* flags, uint32_t n, int32_t want, uint32_t max_polls, int32_t* timed_out) {
    for (uint32_t r = threadIdx.x; r < n; r += 64) {
        uint32_t polls = 0;
        while (__hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
            if (++polls >= max_polls) {
                __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(64);
        }
    }
}

extern "C" int lz_wait_flags(const int32_t* flags, uint32_t n, int32_t want, uint32_t max_polls, int32_t* timed_out, lz_stream_t stream) {
    LZ_REQUIRE(flags && timed_out, LZ_ERR_BAD_ARGUMENT, "wait_flags: null");
    LZ_REQUIRE(n >= 1 && n <= 4096 && max_polls >= 1, LZ_ERR_BAD_ARGUMENT, "wait_flags: 1 <= n <= 4096, max_polls >= 1");
    hipLaunchKernelGGL(lz_k_wait_flags, dim3(1), dim3(64), 0, lz_st(stream), flags, n, want, max_polls, timed_out);
    LZ_CHECK_LAUNCH("wait_flags");
    return LZ_OK;
}

extern "C" int lz_timing_destroy(lz_timing* t) {
    if (!t) return LZ_OK;
    for (auto& e : t->ev) (void)hipEventDestroy(e);
    delete t;
    return LZ_OK;
}

extern "C" int lz_timing_reset(lz_timing* t) {
    LZ_REQUIRE(t, LZ_ERR_BAD_ARGUMENT, "timing_reset: null");
    t->used = 0;
    return LZ_OK;
}

// record the begin (which = 0) / end (which = 1) event of the next pair on `stream`: callers that bracket something other than the
// loop's head launches (the fused frame kernel) use the same lz_timing object
extern "C" int lz_timing_mark(lz_timing* t, int which, lz_stream_t stream) {
    LZ_REQUIRE(t, LZ_ERR_BAD_ARGUMENT, "timing_mark: null");
    if ((size_t)(t->used + 1) * 2 > t->ev.size()) return LZ_OK;   // full: further pairs are dropped
    hipError_t rc = hipEventRecord(t->ev[2 * t->used + (which ? 1 : 0)], lz_st(stream));
    if (rc != hipSuccess) { lz_set_error("timing_mark: %s", hipGetErrorString(rc)); return (int)rc; }
    if (which) t->used++;
    return LZ_OK;
}

// elapsed time of every recorded pair, in ms; call after the stream has been synchronised.  Returns the pair count.
extern "C" int lz_timing_elapsed_ms(lz_timing* t, float* out_ms, uint32_t capacity, uint32_t* n_pairs) {
    LZ_REQUIRE(t && out_ms && n_pairs, LZ_ERR_BAD_ARGUMENT, "timing_elapsed_ms: null argument");
    const uint32_t n = t->used < capacity ? t->used : capacity;
    for (uint32_t i = 0; i < n; i++) {
        hipError_t rc = hipEventElapsedTime(&out_ms[i], t->ev[2 * i], t->ev[2 * i + 1]);
        if (rc != hipSuccess) {
            lz_set_error("timing_elapsed_ms: pair %u: %s", i, hipGetErrorString(rc));
            return (int)rc;
        }
    }
    *n_pairs = n;
    return LZ_OK;
}

extern "C" int lz_loop_run(const lz_frame* f, uint32_t parity, uint32_t n_iterations, lz_timing* timing, lz_stream_t stream) {
    LZ_REQUIRE(f, LZ_ERR_BAD_ARGUMENT, "loop_run: null");
    if (f->N == 0) return LZ_OK;                 // no ray: nothing to enqueue (the per-ray arrays of an empty batch have no storage)
    LZ_REQUIRE(f->state && f->workspace && f->rays_alive[0] && f->rays_alive[1], LZ_ERR_BAD_ARGUMENT, "loop_run: incomplete lz_frame");
    uint32_t cur = parity & 1u;
    const int32_t* count = reinterpret_cast<const int32_t*>(f->state) + LZ_LOOP_NEXT + 2;  // n_samples of the iteration in flight
    const uint32_t rows = f->sample_budget > f->N ? f->sample_budget : f->N;  // capacity of the sample buffers
    for (uint32_t it = 0; it < n_iterations; it++) {
        const uint32_t nxt = cur ^ 1u;
        int rc = lz_loop_march(f->state, f->N, f->sample_budget, f->n_step_cap, f->rays_alive[cur], f->rays_alive[nxt], f->workspace, f->rays_t,
                               f->rays_o, f->rays_d, f->bound, f->dt_gamma, f->max_steps, f->C, f->H, f->grid, f->nears, f->fars, f->xyzs,
                               f->dirs, f->deltas, f->ray_counts, stream);
        if (rc != LZ_OK) return rc;
        const bool timed = timing && (size_t)(timing->used + 1) * 2 <= timing->ev.size();
        if (timed) (void)hipEventRecord(timing->ev[2 * timing->used], lz_st(stream));
        rc = lz_triplane_head_forward(&f->head, f->xyzs, f->dirs, rows, count, f->sigmas, f->rgbs, f->amb_aud, f->amb_eye, f->unc, stream);
        if (rc != LZ_OK) return rc;
        if (timed) {
            (void)hipEventRecord(timing->ev[2 * timing->used + 1], lz_st(stream));
            timing->used++;
        }
        rc = lz_loop_composite(f->state, f->N, f->T_thresh, f->rays_alive[nxt], f->rays_t, f->sigmas, f->rgbs, f->deltas, f->amb_aud,
                               f->amb_eye, f->unc, f->weights_sum, f->depth, f->image, f->amb_aud_sum, f->amb_eye_sum, f->unc_sum,
                               f->workspace, stream);
        if (rc != LZ_OK) return rc;
        cur = nxt;
    }
    return LZ_OK;
}
