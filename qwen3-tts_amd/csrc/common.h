// common.h -- shared device/host helpers for libqtts (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <stdexcept>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <thread>
#include <functional>

#include "../../include/qtts.h"

namespace qtts {

// ------------------------------------------------------------------------------------------ errors
void set_last_error(const std::string& s);

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define QTTS_CHECK_HIP(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            throw ::qtts::Error(QTTS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define QTTS_REQUIRE(cond, code, msg)                                  \
    do {                                                               \
        if (!(cond)) throw ::qtts::Error((code), std::string(msg));    \
    } while (0)

// ------------------------------------------------------------------------------------------ A/B switches
// One table for every A/B switch of the library (include/qtts.h: qtts_set_option).  QTTS_ENV("QTTS_X") = the value qtts_set_option last
// gave the switch, else the environment variable of that name, else null.  The lookup is cached per call site and thread and repeated
// only when the table's generation has moved (one relaxed atomic load per use: nothing a launch path notices).
// Engine-level switches are copied into the handle when it is created; launcher-level switches follow the table at once.  Tests
// flip switches through the C ABI -- never through os.environ inside a process.
const char* opt_lookup(const char* var, unsigned& gen_seen, std::string& cache, bool& has);     // (codec_engine.hip)
#define QTTS_ENV(var)                                                                  \
    ([]() -> const char* {                                                            \
        static thread_local unsigned gen_ = 0;                                        \
        static thread_local std::string val_;                                         \
        static thread_local bool has_ = false;                                        \
        return ::qtts::opt_lookup(var, gen_, val_, has_);                             \
    }())
// integer / boolean forms: QTTS_OPT_INT("QTTS_X", default), QTTS_OPT_ON("QTTS_X") (unset = on), QTTS_OPT_SET("QTTS_X") (unset = off)
#define QTTS_OPT_INT(var, dflt) ([&]() -> int { const char* e_ = QTTS_ENV(var); return e_ ? atoi(e_) : (dflt); }())
#define QTTS_OPT_ON(var) ([]() -> bool { const char* e_ = QTTS_ENV(var); return !e_ || atoi(e_) != 0; }())
#define QTTS_OPT_SET(var) ([]() -> bool { const char* e_ = QTTS_ENV(var); return e_ && atoi(e_) != 0; }())

// hipFuncSetAttribute(kern, MaxDynamicSharedMemorySize, bytes) once per (kernel, DEVICE), thread-safe: a process-wide `static bool`
// would leave a second device of the process without the attribute and is shared between engine threads.  (codec_engine.hip)
void ensure_dynamic_lds(const void* kern, int bytes);

// ------------------------------------------------------------------------------------------ bf16
typedef uint16_t bf16_t;

__host__ __device__ inline bf16_t f32_to_bf16(float f) {  // round-to-nearest-even, NaN preserved
    union { float f; uint32_t u; } v;
    v.f = f;
    uint32_t u = v.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__host__ __device__ inline float bf16_to_f32(bf16_t h) {
    union { float f; uint32_t u; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}

typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
__device__ inline unsigned pack_bf16(float a, float b) {   // two floats -> one dword of bf16 (a low, b high): v_cvt_pk_bf16_f32 (RNE)
    hwbf16x2 v = {(__bf16)a, (__bf16)b};
    return *reinterpret_cast<unsigned*>(&v);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));  // 8 bf16 = 4 VGPRs (MFMA operand type)
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ cross-lane reductions
// DPP row rotations (VALU rate) instead of ds_bpermute shuffles (LDS crossbar, ~100 cycles each) for the reductions the
// hot kernels run per key / per row: after ror 8/4/2/1 every lane of a 16-lane row holds the row's sum.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x122>(v);   // row_ror:2
    v += dpp_mov<0x121>(v);   // row_ror:1
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0x128>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x122>(v));
    v = fmaxf(v, dpp_mov<0x121>(v));
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// whole-wave results combined from the four row results in a fixed order, identical in every lane
__device__ __forceinline__ float wave_sum64_dpp(float v) {
    v = row16_sum(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max64_dpp(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each 16-lane row on DPP rotations (what a rotation wraps around
// is masked by the lane index), then the running totals of the rows below via readlane -- ~16 VALU instructions instead of six
// ds_bpermute round trips
__device__ __forceinline__ float wave_incl_scan64_dpp(float v, int lane) {
    const int li = lane & 15, row = lane >> 4;
    float t;
    t = dpp_mov<0x121>(v); v += li >= 1 ? t : 0.f;
    t = dpp_mov<0x122>(v); v += li >= 2 ? t : 0.f;
    t = dpp_mov<0x124>(v); v += li >= 4 ? t : 0.f;
    t = dpp_mov<0x128>(v); v += li >= 8 ? t : 0.f;
    const float r0 = lane_bcast(v, 15), r1 = lane_bcast(v, 31), r2 = lane_bcast(v, 47);
    return v + (row == 0 ? 0.f : (row == 1 ? r0 : (row == 2 ? r0 + r1 : (r0 + r1) + r2)));
}


// ------------------------------------------------------------------------------------------ device memory
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        QTTS_CHECK_HIP(hipMalloc(&p, n));
        bytes = n;
    }
    void ensure(size_t n) { if (n > bytes) alloc(n); }
    void upload(const void* host, size_t n) {
        ensure(n);
        QTTS_CHECK_HIP(hipMemcpy(p, host, n, hipMemcpyHostToDevice));
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// host-side source tensor view (bind)
struct HostTensor {
    const void* data;
    int dtype;  // QTTS_F32 | QTTS_BF16
    std::vector<int64_t> shape;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
    float at(int64_t i) const {
        return dtype == QTTS_F32 ? ((const float*)data)[i] : bf16_to_f32(((const bf16_t*)data)[i]);
    }
    std::vector<float> to_f32() const {
        std::vector<float> v((size_t)numel());
        if (dtype == QTTS_F32) memcpy(v.data(), data, v.size() * 4);
        else for (size_t i = 0; i < v.size(); ++i) v[i] = bf16_to_f32(((const bf16_t*)data)[i]);
        return v;
    }
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// host-side helper for the one-time weight repacking at bind/finalize
inline void parallel_for(int64_t n, const std::function<void(int64_t, int64_t)>& fn) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 32) nt = 32;
    if (n < 64 || nt == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    const int64_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const int64_t a = t * chunk, b = std::min<int64_t>(n, a + chunk);
        if (a >= b) break;
        th.emplace_back([=, &fn] { fn(a, b); });
    }
    for (auto& x : th) x.join();
}

}  // namespace qtts
