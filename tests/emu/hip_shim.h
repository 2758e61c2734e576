/* hip_shim.h -- X3_EMU builds only (tests): the handful of HIP runtime calls api.hip makes, on plain host memory.
 * Device allocations are filled with 0xA5 so that any reliance on zero-initialised workspace shows up. */
#ifndef X3_HIP_SHIM_H
#define X3_HIP_SHIM_H
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef void *hipStream_t;
struct x3emu_event { double t; };
typedef x3emu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (!*p) return hipErrorOutOfMemory; memset(*p, 0xA5, n); return hipSuccess; }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
/* X3EMU_DEVICES=N: N emulated devices (they share the host's memory; a handle per device exercises the multi-device code of api.hip) */
static inline hipError_t hipGetDeviceCount(int *n) { const char *e = getenv("X3EMU_DEVICES"); const int v = e ? atoi(e) : 1; *n = v >= 1 && v <= 64 ? v : 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; } /* (the emulated device: the masked streams of the sliced schedule are created like on the real one) */
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (void *)1; return hipSuccess; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, unsigned, const unsigned *) { *s = (void *)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new x3emu_event(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); e->t = t.tv_sec * 1e3 + t.tv_nsec * 1e-6; return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
enum { hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
#endif
