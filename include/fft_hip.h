/*
 * fft_hip.h -- C ABI of the MI355X (gfx950) HIP FFT backend, and the
 * additive public entry points built on it.
 *
 * Part 1 is the drop-in boundary: the 13 backend functions the reference's
 * dispatcher expects of a backend -- the `extern` block of gpu/fft_gpu.c:32-46
 * (CUDA flavour; the functions being replaced live in gpu/fft_cuda.cu:53-252).
 * Plain C types only; opaque handles; no torch / HIP types in signatures
 * (streams travel as void*).  INTEGRATION.md shows the three-line patch that
 * binds the reference's own fft_gpu.c to these symbols.
 *
 * Part 2 is additive (nothing in the reference has these): fp32, batches with
 * 64-bit offsets, raw device pointers, streams, multi-device, algorithm
 * selection, the stand-alone bit-reversal permutation, HIP-event timing.
 */
#ifndef FFT_HIP_H
#define FFT_HIP_H

#include <stdint.h>
#include "fft_common.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifndef FFT_GPU_H
typedef struct fft_gpu_memory* fft_gpu_memory_t;
typedef struct fft_gpu_plan* fft_gpu_plan_t;
#endif

/* ------------------------------------------------------------------ part 1
 * Backend symbol set (replaces fft_gpu_*_cuda, gpu/fft_cuda.cu).          */
int fft_gpu_init_hip(void);               /* replaces fft_gpu_init_cuda        fft_cuda.cu:53-83   */
void fft_gpu_cleanup_hip(void);           /* replaces fft_gpu_cleanup_cuda     fft_cuda.cu:86-91   */
int fft_gpu_available_hip(void);          /* replaces fft_gpu_available_cuda   fft_cuda.cu:94-100  */
fft_gpu_memory_t fft_gpu_alloc_hip(size_t n_complex);                 /* fft_cuda.cu:103-115 */
void fft_gpu_free_hip(fft_gpu_memory_t mem);                          /* fft_cuda.cu:118-123 */
void fft_gpu_copy_h2d_hip(fft_gpu_memory_t dst, const complex_t* src, size_t n_complex); /* :126-129 */
void fft_gpu_copy_d2h_hip(complex_t* dst, fft_gpu_memory_t src, size_t n_complex);       /* :132-135 */
fft_gpu_plan_t fft_gpu_plan_1d_hip(int n, int batch, fft_direction dir);                 /* :138-163 */
/* `dir` is accepted for signature compatibility and IGNORED: the reference's
 * dispatcher hard-codes FFT_FORWARD there (gpu/fft_gpu.c:252); the plan's own
 * direction is used. */
void fft_gpu_execute_hip(fft_gpu_plan_t plan, fft_gpu_memory_t in, fft_gpu_memory_t out, fft_direction dir); /* :166-185 */
void fft_gpu_destroy_plan_hip(fft_gpu_plan_t plan);                   /* fft_cuda.cu:188-193 */
const char* fft_gpu_get_device_name_hip(void);                        /* fft_cuda.cu:196-206 */
void fft_gpu_get_memory_info_hip(size_t* total, size_t* available);   /* fft_cuda.cu:209-211 */
int fft_gpu_dft_1d_hip(complex_t* in, complex_t* out, int n, fft_direction dir); /* fft_cuda.cu:214-252 */

/* ------------------------------------------------------------------ part 2 */
typedef enum {
    FFT_PREC_F64 = 0, /* complex_t   : interleaved double (the reference's only type) */
    FFT_PREC_F32 = 1  /* complex32_t : interleaved float  (additive)                  */
} fft_precision_t;

/* Which butterfly family the power-of-two engine uses.  All compute the same
 * DFT (as radix4_fft / split_radix_fft / radix2_dit_fft do in the reference:
 * algorithms/core/radix4.c:98-125, split_radix.c:23-55 are radix-2 loops). */
typedef enum {
    FFT_GPU_ALGO_AUTO = 0,          /* fastest known schedule for the size               */
    FFT_GPU_ALGO_RADIX2 = 1,        /* LDS Stockham, radix-2 passes only                 */
    FFT_GPU_ALGO_RADIX4 = 2,        /* LDS Stockham, radix-4 passes (+ one radix-2)      */
    FFT_GPU_ALGO_SPLIT_RADIX = 3,   /* LDS Stockham, radix-8/16 passes with split-radix (L-shaped) codelets */
    FFT_GPU_ALGO_RADIX2_GLOBAL = 4, /* reference-shaped: bit-reversal permutation kernel + log2(n)
                                       in-place radix-2 DIT stage kernels in HBM (radix2_dit.c:70-112) */
    FFT_GPU_ALGO_BLUESTEIN = 5,     /* chirp-z even when n is a power of two (bluestein.c:79-155);
                                       every non-power-of-two n uses it whatever algo says */
    FFT_GPU_ALGO_RADIX2_SHFL = 6    /* reference-shaped radix-2 DIT held by one wavefront per transform: LDS
                                       bit-reversal permutation, in-register stages, __shfl_xor for the
                                       cross-lane strides (n = 128..1024; other n: FFT_GPU_ALGO_RADIX2) */
} fft_gpu_algo_t;

typedef struct {
    int n;              /* transform length */
    int batch;          /* transforms per execute */
    int direction;      /* -1 / +1 */
    int precision;      /* fft_precision_t */
    int algo;           /* fft_gpu_algo_t actually used */
    int device;         /* HIP device ordinal the plan lives on */
    int bluestein_m;    /* 0 for power-of-two n, else the padded length */
    int n_passes;       /* HBM round trips of the power-of-two engine (1, 2 or 3; log2n+1 for RADIX2_GLOBAL) */
    int factors[4];     /* length of the LDS-resident sub-transform of each pass */
    int chunk_batch;    /* transforms processed per launch group (Infinity-Cache blocking) */
    size_t workspace_bytes;
    int team_tiles;     /* 0: multi-pass schedule only.  1/2/4: the plan runs the one-round-trip team kernel (a whole
                           transform per XCD, this many 64 KiB tiles per workgroup); the multi-pass schedule
                           described by n_passes/factors is queued behind it as its fallback */
    int fused;          /* Bluestein and fused-consumer plans: 0 the element-wise steps run as kernels of their own, 1 they
                           ride on the first load / last store of the transforms, 2 and the forward transform's last pass and
                           the inverse's first pass are ONE kernel (2 * n_passes - 1 launches per group of chunk_batch), 3 the padded
                           transform fits one LDS tile and forward transform, product and inverse transform are ONE kernel */
    int team_kernel;    /* which one-round-trip kernel team_tiles refers to: 0 none, 1 team_fft_kernel (csrc/fft_team.h), 2
                           team_defer_kernel (csrc/fft_team_defer.h), 3 team_quad_kernel (csrc/fft_team_quad.h: whole-line row
                           segments, both steps decimated by 4, the exchange in four rounds through the XCD's L2) */
} fft_gpu_plan_info_t;

/* Per-plan switches (tests and integrators; nothing here changes results) */
typedef enum {
    FFT_GPU_OPT_TEAM_FORCE_FALLBACK = 1, /* 1: the team kernel behaves as if it could not form its XCD teams (status 1, nothing
                                            touched) and the multi-pass plan queued behind it does the work */
    FFT_GPU_OPT_TEAM_ENABLE = 2,         /* 0: run the multi-pass schedule only; 1: back to the team kernel where the plan has one */
    FFT_GPU_OPT_NO_FUSION = 3,           /* 1: Bluestein / fused-consumer plans run their element-wise steps as kernels of their own
                                            instead of fusing them into the FFT passes (same results to rounding; tests) */
    FFT_GPU_OPT_NO_CHAIN = 4,            /* 1: Bluestein / fused-consumer plans keep the forward transform's last pass and the inverse
                                            transform's first pass as two kernels (by default they run as one where their tiles agree
                                            and the transform has >= 2^19 points; tests); 2: as one wherever the tiles agree (tools) */
    FFT_GPU_OPT_TEAM_NO_REPLAY = 5       /* 1: the caller does NOT keep the buffers of its asynchronous executes alive and unmodified until
                                            the next fft_gpu_plan_sync_hip (e.g. they come from a stream-ordered caching allocator): a team
                                            kernel timeout is then never repaired by repeating executes, fft_gpu_plan_sync_hip returns -1 */
} fft_gpu_plan_option_t;

/* Fused consumers of the transform (reference applications/convolution.c, applications/power_spectrum.c): FFT ->
 * element-wise -> inverse FFT with the zero padding, the spectral product and the truncation fused into the FFT passes. */
typedef enum {
    FFT_GPU_FUSED_CONV_LINEAR = 0,   /* y = x * h, x: nx, h: nh (fixed at plan time), y: nx + nh - 1   convolution.c:34-69   */
    FFT_GPU_FUSED_CONV_CIRCULAR = 1, /* y = x (*) h, all of length nx (a power of two)                   convolution.c:72-96   */
    FFT_GPU_FUSED_AUTOCORR = 2,      /* first nx lags of IFFT(|FFT(x, zero padded)|^2)                   power_spectrum.c:133-158 */
    FFT_GPU_FUSED_XCORR = 3,         /* first nx lags of IFFT(conj(FFT(x)) FFT(y))                       power_spectrum.c:161-190 */
    FFT_GPU_FUSED_PSD = 4            /* one-sided Hann periodogram, nx/2 + 1 REAL values (nx a power of two) power_spectrum.c:58-86 */
} fft_gpu_fused_t;

/* backend-level additive entry points */
int fft_gpu_device_count_hip(void);
int fft_gpu_set_device_hip(int device);
int fft_gpu_get_device_hip(void);
fft_gpu_memory_t fft_gpu_alloc_bytes_hip(size_t bytes);
int fft_gpu_copy_h2d_bytes_hip(fft_gpu_memory_t dst, const void* src, size_t bytes);
int fft_gpu_copy_d2h_bytes_hip(void* dst, fft_gpu_memory_t src, size_t bytes);
void* fft_gpu_memory_ptr_hip(fft_gpu_memory_t mem);
size_t fft_gpu_memory_bytes_hip(fft_gpu_memory_t mem);
/* page-lock / release a host array for the lifetime of a plan that borrows it (pinned H2D / D2H); 0 / -1 */
int fft_gpu_host_register_hip(void* host_ptr, size_t bytes);
int fft_gpu_host_unregister_hip(void* host_ptr);
int fft_gpu_host_is_registered_hip(const void* host_ptr); /* 1 page-locked and known to the runtime, 0 not */
/* the box's practical memory ceiling: best read + write rate (GB/s) of a plain 16-byte-per-lane device copy over `bytes`
 * bytes, a few launch shapes, `iters` launches each; -1 on failure */
double fft_gpu_copy_bench_hip(size_t bytes, int iters);
/* the same yardstick per direction: mode 0 copy (read + written bytes / s), 1 read-only stream, 2 write-only stream; the best of
 * several launch shapes with and without the non-temporal hint (copy: also an LDS-DMA tile copy in the engine's own shape), GB/s */
double fft_gpu_stream_bench_hip(size_t bytes, int iters, int mode);
/* resource counters of this process (tests): device allocations and streams the backend has created so far */
void fft_gpu_debug_counters_hip(long long* device_allocations, long long* streams_created);
/* FFT_MEASURE at the device level: time the plan's candidate schedules (team kernel vs multi-pass) on scratch buffers of
 * the plan's own size and keep the faster; returns 1 team kernel kept, 0 multi-pass kept / nothing to choose, -1 error */
int fft_gpu_plan_measure_hip(fft_gpu_plan_t plan, int iters);
fft_gpu_plan_t fft_gpu_plan_1d_ex_hip(int n, int batch, fft_direction dir, fft_precision_t prec, fft_gpu_algo_t algo);
/* 2D complex transforms of `n_matrices` row-major rows x cols matrices: rows = one batched 1D execute, columns = the
 * strided column pass of the four-step engine (or transpose + batched 1D); the inverse is scaled once by 1/(rows*cols).
 * Execute with fft_gpu_execute(_ptr/_async); in == out allowed.  fft_gpu_plan_2d_hip replaces the stub gpu/fft_gpu.c:377-385. */
fft_gpu_plan_t fft_gpu_plan_2d_hip(int rows, int cols, fft_direction dir);
fft_gpu_plan_t fft_gpu_plan_2d_ex_hip(int rows, int cols, int n_matrices, fft_direction dir, fft_precision_t prec);
/* real-input forward / real-output inverse 1D transforms (reference stubs algorithms/auto/fft_auto.c:391-409): r2c reads
 * [batch][n] reals and writes [batch][n/2 + 1] complex bins; c2r the reverse, scaled by 1/n.  Execute with fft_gpu_execute_ptr. */
fft_gpu_plan_t fft_gpu_plan_r2c_1d_hip(int n, int batch, fft_precision_t prec);
fft_gpu_plan_t fft_gpu_plan_c2r_1d_hip(int n, int batch, fft_precision_t prec);
/* fused consumers; h_host: the nh kernel samples (host memory, complex of `prec`) of the two convolutions, else ignored */
fft_gpu_plan_t fft_gpu_plan_fused_hip(fft_gpu_fused_t kind, int nx, int nh, const void* h_host, int batch, fft_precision_t prec);
int fft_gpu_fused_out_len_hip(fft_gpu_plan_t plan); /* elements per output row (complex; FFT_GPU_FUSED_PSD: real) */
/* async on the plan's stream.  d_x: [batch][nx] complex; d_y: the second signal of FFT_GPU_FUSED_XCORR, else NULL;
 * d_out: [batch][out_len]; sample_rate scales FFT_GPU_FUSED_PSD only */
int fft_gpu_execute_fused_hip(fft_gpu_plan_t plan, const void* d_x, const void* d_y, void* d_out, double sample_rate);
int fft_gpu_plan_info_hip(fft_gpu_plan_t plan, fft_gpu_plan_info_t* info);
/* NULL = the plan's own (non-blocking) stream.  A caller that works on HIP's default stream -- PyTorch's default stream
 * has the handle 0 -- names it explicitly: (void*)1 = hipStreamLegacy, or its work and the plan's are not ordered. */
int fft_gpu_plan_set_stream_hip(fft_gpu_plan_t plan, void* hip_stream);
int fft_gpu_execute_ptr_hip(fft_gpu_plan_t plan, const void* d_in, void* d_out); /* async on the plan's stream */
/* Waits for the plan's stream.  0: every execute since the last sync holds valid results.  Should a team kernel's bounded wait
 * have run out (a member of a formed team stopped making progress: a hardware fault, not load -- formation under load falls back
 * BEFORE anything is touched), the team kernel is retired for this plan and the executes since the last sync are repeated on the
 * multi-pass schedule before this returns (still 0) -- WHERE THAT IS PROVABLY RIGHT: the plan is a plain 1D complex plan, and the
 * execute's input still holds the caller's data, i.e. it is out of place and no LATER execute since the last sync wrote into its
 * input (A -> B then B -> A: the second launch has overwritten A), or it is in place and ran from the plan's staged copy (the plan
 * stages the input of in-place executes until one sync has seen its team kernel end well, so the FIRST use of a plan on a device is
 * always repairable).  Contract for asynchronous callers: every buffer handed to fft_gpu_execute_ptr_hip stays allocated and
 * unmodified until the next sync (or set FFT_GPU_OPT_TEAM_NO_REPLAY).  -1: the stream failed, or a timeout hit an execute that
 * could not be repeated (stderr says how many): that data is invalid.  Bluestein, 2D, real and fused plans never repeat (their
 * cores transform the plan's own intermediates): a timeout in one of their cores is reported as -1.  (fft_gpu_execute() of
 * fft_gpu.h is void, as in the reference: gpu/fft_cuda.cu:166-185, and blocks: its buffers are the caller's for the duration.) */
int fft_gpu_plan_sync_hip(fft_gpu_plan_t plan);
int fft_gpu_plan_set_option_hip(fft_gpu_plan_t plan, fft_gpu_plan_option_t option, int value);
/* planner policy for plans created after the call; a negative argument keeps the current value.  team_mode: 0 never
 * the team kernel, 1 (default) where it measured faster than the multi-pass schedule, 2 every size it is built for;
 * team_min_batch: 0 = the measured batch crossover; chunk_mb: 0 = the default multi-pass launch-group size.
 * fft_gpu_init seeds them once from FFT_HIP_TEAM / FFT_HIP_TEAM_MIN_BATCH / FFT_HIP_CHUNK_MB. */
int fft_gpu_set_policy_hip(int team_mode, int team_min_batch, int chunk_mb);
/* syncs the plan's stream, then: 0 the last execute was done by the team kernel, 1 its XCD teams could not be
 * formed and the multi-pass fallback did the work, 2 a team barrier timed out (results invalid; plan_sync returns -1
 * too), -1 the plan has no team kernel or it has never been launched */
int fft_gpu_plan_team_status_hip(fft_gpu_plan_t plan);
/* profiling: every workgroup of the team kernel logs its 100 MHz clock at its first `events` timeline events into
 * d_trace[workgroup * events + i] (256 * events * 8 bytes of device memory owned by the caller); NULL switches it off.
 * team_quad_kernel keeps the last two slots for itself: [events - 2] = the workgroup's clock at kernel entry, [events - 1] =
 * (team << 8) | seat (tools/quad_trace.py) */
int fft_gpu_plan_team_trace_hip(fft_gpu_plan_t plan, void* d_trace, int events);
/* `iters` back-to-back executes bracketed by hipEvents recorded on the plan's stream */
int fft_gpu_execute_timed_hip(fft_gpu_plan_t plan, const void* d_in, void* d_out, int iters, float* elapsed_ms);
/* one execute with a HIP event after every pass launch: per-pass device milliseconds and launch counts */
int fft_gpu_profile_passes_hip(fft_gpu_plan_t plan, const void* d_in, void* d_out, float* ms, int* launches, int max_passes);
int fft_gpu_dft_1d_batch_hip(const void* in, void* out, int n, int batch, fft_direction dir, fft_precision_t prec);
/* out[b][bit_reverse(i)] = in[b][i]; in == out allowed (reference: the swap loop radix2_dit.c:70-77) */
int fft_gpu_bit_reverse_hip(const void* d_in, void* d_out, int n, int batch, fft_precision_t prec, void* hip_stream);

/* public additive API (dispatcher level; same style as fft_gpu.h) */
int fft_gpu_device_count(void);
fft_gpu_memory_t fft_gpu_alloc_f32(size_t n_complex32);
void fft_gpu_copy_h2d_f32(fft_gpu_memory_t dst, const complex32_t* src, size_t n);
void fft_gpu_copy_d2h_f32(complex32_t* dst, fft_gpu_memory_t src, size_t n);
void* fft_gpu_memory_ptr(fft_gpu_memory_t mem);
fft_gpu_plan_t fft_gpu_plan_1d_f32(int n, int batch, fft_direction direction);
fft_gpu_plan_t fft_gpu_plan_1d_ex(int n, int batch, fft_direction direction, fft_precision_t prec, fft_gpu_algo_t algo);
int fft_gpu_plan_info(fft_gpu_plan_t plan, fft_gpu_plan_info_t* info);
int fft_gpu_plan_set_stream(fft_gpu_plan_t plan, void* hip_stream);
int fft_gpu_execute_async(fft_gpu_plan_t plan, fft_gpu_memory_t in, fft_gpu_memory_t out);
int fft_gpu_execute_ptr(fft_gpu_plan_t plan, const void* d_in, void* d_out);
int fft_gpu_plan_sync(fft_gpu_plan_t plan);
int fft_gpu_execute_timed(fft_gpu_plan_t plan, const void* d_in, void* d_out, int iters, float* elapsed_ms);
int fft_gpu_dft_1d_f32(complex32_t* in, complex32_t* out, int n, fft_direction direction);
int fft_gpu_dft_1d_batch_f32(complex32_t* in, complex32_t* out, int n, int batch, fft_direction direction);
int fft_gpu_bit_reverse(fft_gpu_memory_t in, fft_gpu_memory_t out, int n, int batch, fft_precision_t prec);

#ifdef __cplusplus
}
#endif

#endif /* FFT_HIP_H */
