/*
 * fft_gpu.h -- backend-agnostic GPU FFT API.
 *
 * Same names, argument meaning, return conventions and enum values as the
 * reference's include/fft_gpu.h:14-177; the dispatcher behind it
 * (fft-implementation-in-c_amd/host/fft_gpu.c) routes to the hand-written
 * HIP engine (backend id FFT_GPU_HIP, additive).  The additive entry points
 * (fp32, raw device pointers, streams, device count, algorithm selection,
 * stand-alone bit-reversal permutation) are declared in fft_hip.h.
 */
#ifndef FFT_GPU_H
#define FFT_GPU_H

#include "fft_common.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FFT_GPU_NONE = 0,
    FFT_GPU_CUDA = 1,   /* kept for source compatibility; never available here */
    FFT_GPU_METAL = 2,  /* kept for source compatibility; never available here */
    FFT_GPU_OPENCL = 3, /* kept for source compatibility; never available here */
    FFT_GPU_HIP = 4,    /* additive: AMD CDNA4 (MI355X, gfx950) hand-written HIP engine */
    FFT_GPU_AUTO = -1
} fft_gpu_backend_t;

typedef struct fft_gpu_memory* fft_gpu_memory_t; /* opaque device buffer handle */
typedef struct fft_gpu_plan* fft_gpu_plan_t;     /* opaque plan handle */

/* lifecycle (reference fft_gpu.h:35-52) */
int fft_gpu_init(fft_gpu_backend_t backend); /* 0 ok, -1 no usable backend; idempotent */
void fft_gpu_cleanup(void);
int fft_gpu_available(void);                 /* 1/0, callable before init */
fft_gpu_backend_t fft_gpu_get_backend(void);

/* device buffers of `size` complex_t elements (reference fft_gpu.h:61-83) */
fft_gpu_memory_t fft_gpu_alloc(size_t size); /* NULL on failure / not initialised */
void fft_gpu_free(fft_gpu_memory_t mem);     /* NULL-safe */
void fft_gpu_copy_h2d(fft_gpu_memory_t dst, const complex_t* src, size_t size);
void fft_gpu_copy_d2h(complex_t* dst, fft_gpu_memory_t src, size_t size);

/* plans (reference fft_gpu.h:94-108).  Batched layout: transform b occupies
 * elements [b*n, (b+1)*n) -- stride 1, distance n (cufftPlanMany call in the
 * reference's gpu/fft_cuda.cu:152-156).  The direction is stored in the plan;
 * the inverse is scaled by 1/n like every CPU algorithm of the reference
 * (algorithms/core/radix2_dit.c:115-119).  n need not be a power of two:
 * other sizes are planned as Bluestein over the power-of-two engine. */
fft_gpu_plan_t fft_gpu_plan_1d(int n, int batch, fft_direction direction);
void fft_gpu_execute(fft_gpu_plan_t plan, fft_gpu_memory_t in, fft_gpu_memory_t out); /* in==out: in place; blocks */
void fft_gpu_destroy_plan(fft_gpu_plan_t plan); /* NULL-safe */

/* host-pointer conveniences (reference fft_gpu.h:120-131): 0 / -1 */
int fft_gpu_dft_1d(complex_t* in, complex_t* out, int n, fft_direction direction);
int fft_gpu_dft_1d_batch(complex_t* in, complex_t* out, int n, int batch, fft_direction direction);

/* 2D complex transforms of one row-major rows x cols matrix (stubs in the reference, gpu/fft_gpu.c:377-394: NULL / -1; real here: rows as
 * one batched 1D execute, columns as the engine's strided column pass, the inverse scaled once by 1 / (rows * cols) -- DESIGN.md 4.5) */
fft_gpu_plan_t fft_gpu_plan_2d(int rows, int cols, fft_direction direction);
int fft_gpu_dft_2d(complex_t* in, complex_t* out, int rows, int cols, fft_direction direction);

/* information (reference fft_gpu.h:163-177) */
const char* fft_gpu_get_device_name(void); /* "No GPU" when none */
void fft_gpu_get_memory_info(size_t* total, size_t* available);
int fft_gpu_set_device(int device);        /* 0 / -1; real here (a stub in the reference) */

#ifdef __cplusplus
}
#endif

#include "fft_hip.h"

#endif /* FFT_GPU_H */
