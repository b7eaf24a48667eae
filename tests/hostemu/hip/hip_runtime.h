// TEST INFRASTRUCTURE: a host stand-in for <hip/hip_runtime.h>, just enough to compile csrc/*.hip -- engines and kernels
// -- as plain C++ with "device" memory = host memory and kernels executed by the SIMT emulator (../simt.h), so that the
// CPU test-suite runs the product's real finalize() repacking, buffer rotation, carry bookkeeping, C ABI and kernel code.
// Never part of the product build.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
static const hipError_t hipSuccess = 0;
typedef void* hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
#define __host__
#define __device__
#define __forceinline__ inline
// Fresh device memory holds garbage on the GPU (the caching allocator recycles blocks): fill it with 0xFF bytes -- NaN
// as fp32 / bf16, -1 as an integer -- so that code relying on zero-initialised buffers fails here too.
inline hipError_t hipMalloc(void** p, size_t n) {
    *p = std::malloc(n ? n : 1);
    if (*p) std::memset(*p, 0xFF, n ? n : 1);
    return *p ? 0 : 2;
}
inline hipError_t hipFree(void* p) { std::free(p); return 0; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return 0; }     // (an MI355X's CU count: the tile chooser's test pins its picks)
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "host emulation"; }

// ---- kernel execution: a SIMT emulator (fibers + wave collectives + workgroup barriers), see ../simt.h.  The build
// script rewrites `extern __shared__` -> `extern` (dynamic LDS = the host arrays defined in lds_arrays.cpp) and
// `__shared__` -> `static` (one workgroup runs at a time) in the copies of the kernel sources it compiles.
#include <cmath>
#define __global__
#define __launch_bounds__(...)
struct alignas(16) float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct alignas(8) float2 { float x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(8) ushort4 { unsigned short x, y, z, w; };
#include "../simt.h"
// ---- copies and synchronisation, capture-aware: an asynchronous copy issued while a stream capture is open becomes a
// graph node (pointers baked, bytes moved at replay); a synchronising call inside a capture is an error on the device
// (hipErrorStreamCaptureUnsupported = 900) and here.
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    if (simt::capture().open) { simt::capture().open->nodes.push_back([=] { std::memmove(d, s, n); }); return 0; }
    std::memmove(d, s, n);
    return 0;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    if (simt::capture().open) return 900;
    std::memcpy(d, s, n);
    return 0;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    if (simt::capture().open) { simt::capture().open->nodes.push_back([=] { std::memset(d, v, n); }); return 0; }
    std::memset(d, v, n);
    return 0;
}
inline hipError_t hipMemset(void* d, int v, size_t n) {
    if (simt::capture().open) return 900;
    std::memset(d, v, n);
    return 0;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return simt::capture().open ? 900 : 0; }
inline hipError_t hipDeviceSynchronize() { return simt::capture().open ? 900 : 0; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return 0; }

// ---- the rest of the runtime surface csrc/talker_engine.hip touches: events (no clock: elapsed time is a constant) and
// stream capture / graphs (recorded launches replayed in order, see simt::Graph).
typedef void* hipEvent_t;
typedef simt::Graph* hipGraph_t;
typedef simt::Graph* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static const unsigned hipStreamNonBlocking = 1;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.f; return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone, hipStreamCaptureStatusActive, hipStreamCaptureStatusInvalidated };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = simt::capture().open ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone; return 0; }
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
    if (simt::capture().open) return 900;
    simt::capture().open = new simt::Graph();
    return 0;
}
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
    simt::Graph* gr = simt::capture().open;
    simt::capture().open = nullptr;
    if (g) *g = gr; else delete gr;
    return gr ? 0 : 901;                                              // hipErrorIllegalState: no capture was open
}
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) {
    if (!g) return 1;
    *e = new simt::Graph(*g);                                         // the executable graph is its own copy of the nodes
    return 0;
}
inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
    if (!e) return 1;
    if (simt::capture().open) return 900;
    for (auto& n : e->nodes) n();
    return 0;
}
inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return 0; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return 0; }
inline hipError_t hipGraphGetNodes(hipGraph_t g, hipGraphNode_t*, size_t* n) { if (n) *n = g ? g->nodes.size() : 0; return 0; }
