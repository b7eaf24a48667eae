// TEST INFRASTRUCTURE: a host stand-in for <hip/hip_runtime.h>, just enough to compile the ENGINE orchestration files
// (csrc/codec_engine.hip, csrc/encoder_engine.hip) as plain C++ with "device" memory = host memory.  Together with
// tests/hostemu/cpu_kernels.cpp (CPU versions of the launch_* interfaces) it lets the CPU test-suite execute the real
// finalize() repacking, buffer rotation, carry bookkeeping and C ABI of those engines.  Never part of the product build.
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
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { std::free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return 0; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "host emulation"; }
