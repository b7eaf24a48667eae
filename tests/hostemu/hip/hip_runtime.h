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

// ---- sequential kernel interpreter: enough of the HIP kernel language to EXECUTE thread-independent kernels (no LDS,
// no __syncthreads, no cross-lane ops) on the host.  hipLaunchKernelGGL walks the grid and the block one thread at a
// time with threadIdx / blockIdx set, so the real kernel sources (stream_kernels.hip, encoder_kernels.hip,
// speaker_kernels.hip) run unmodified -- their index arithmetic is what the CPU suite then checks.
#include <cmath>
#define __global__
#define __launch_bounds__(n)
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline dim3 threadIdx, blockIdx, blockDim, gridDim;
template <class F>
inline void qtts_hostemu_launch(dim3 grid, dim3 block, F&& body) {
    gridDim = grid; blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx)
                for (unsigned tx = 0; tx < block.x; ++tx) {
                    blockIdx = dim3(bx, by, bz); threadIdx = dim3(tx, 0, 0);
                    body();
                }
}
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    qtts_hostemu_launch((grid), (block), [&] { kern(__VA_ARGS__); })

// ---- the rest of the runtime surface csrc/talker_engine.hip touches.  Stream capture is not emulated: the talker is run
// with use_graph = 0 here (its eager path launches exactly the kernels a captured frame replays).
#define __shared__ static
inline void __syncthreads() {}
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static const unsigned hipStreamNonBlocking = 1;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.f; return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return 801; }       // hipErrorNotSupported
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { if (g) *g = nullptr; return 801; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return 801; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 801; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return 0; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }
inline hipError_t hipGraphGetNodes(hipGraph_t, hipGraphNode_t*, size_t* n) { if (n) *n = 0; return 0; }
