// granule.h -- hand-off of values BETWEEN WORKGROUPS OF ONE LAUNCH (attention.hip: cp_attn_o_kernel; cp_mlp.hip: cp_mlp_kernel).
// Every value travels as an 8-byte granule {32-bit payload, launch tag}, stored write-through (sc1) and read with sc1 loads until it
// carries the tag of the running launch: no ticket, no fence, and a consumer can tell a stale granule from a fresh one by the tag alone.
// Tag = (frame serial << 7 | launch slot) -- nobody in the launch writes the word it derives from, and the launches that wrote the same
// buffer before had another (serial, slot) pair (the engine keeps slot = position * layers + layer below 128 and refuses the fused
// launches for a code predictor with more than CP_FUSED_MAX_LAYERS layers: talker_engine.hip).
// A consumer cannot hang the device: after GRANULE_SPIN_LIMIT re-reads it gives up -- it raises the engine's flag and latches the
// generation's stop flag (talker_engine.hip: the give-up latch).  What a workgroup that gave up still publishes carries a VALID tag and
// is garbage; nothing consumes it: every later kernel of the chain returns at its `done` check, the host call fails with
// QTTS_ERR_STATE and the Python wrappers re-run the request once on the separate launches (talker.py: TalkerEngine.generate).
#pragma once
#include "common.h"

namespace qtts {
typedef unsigned int cu32x4 __attribute__((ext_vector_type(4)));
#ifdef QTTS_HOST_EMU
struct WtBuf { unsigned char* base; };
__device__ inline WtBuf wt_buf(void* p, size_t) { return WtBuf{static_cast<unsigned char*>(p)}; }
__device__ inline void wt_store16(const WtBuf& b, int off, cu32x4 v) { *reinterpret_cast<cu32x4*>(b.base + off) = v; }
__device__ inline cu32x4 wt_load16(const WtBuf& b, int off) { return *reinterpret_cast<const cu32x4*>(b.base + off); }
__device__ inline cu32x4 wt_load16_cached(const WtBuf& b, int off) { return *reinterpret_cast<const cu32x4*>(b.base + off); }
__device__ inline uint2 wt_load8(const WtBuf& b, int off) { return *reinterpret_cast<const uint2*>(b.base + off); }
__device__ inline void wt_first_pause(int) {}
constexpr int GRANULE_SPIN_LIMIT = 2;                      // (workgroups run one after the other here: a second read never helps)
#else
struct WtBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ WtBuf wt_buf(void* p, size_t bytes) { return WtBuf{__builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000)}; }
// aux = 16: sc1 -- the store writes through to memory, the load is not served from this CU's L1
__device__ __forceinline__ void wt_store16(const WtBuf& b, int off, cu32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, b.r, off, 0, 16); }
__device__ __forceinline__ cu32x4 wt_load16(const WtBuf& b, int off) { return __builtin_amdgcn_raw_buffer_load_b128(b.r, off, 0, 16); }
// aux = 0: an ordinary load -- this CU's L1 and this XCD's L2 may serve it.  Only for granules whose buffer region has not been read
// for a long time (cp_layer.hip rotates its hidden-row regions) AND only behind a fresh sentinel: a line that is stale in a cache comes back
// with an old tag, which the reader sees and answers with sc1 re-reads.
__device__ __forceinline__ cu32x4 wt_load16_cached(const WtBuf& b, int off) { return __builtin_amdgcn_raw_buffer_load_b128(b.r, off, 0, 0); }
__device__ __forceinline__ uint2 wt_load8(const WtBuf& b, int off) {
    typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));
    const cu32x2 v = __builtin_amdgcn_raw_buffer_load_b64(b.r, off, 0, 16);
    uint2 r; r.x = v[0]; r.y = v[1];
    return r;
}
__device__ __forceinline__ void wt_first_pause(int n) { for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1); }        // n x 64 clocks (16 ~ 0.4 us)
constexpr int GRANULE_SPIN_LIMIT = 1 << 18;                // ~0.3 s of re-reads: a producer that never stores is a bug, not a wait
#endif
}  // namespace qtts
