// tstamp.h -- DEBUG build variant `tstamp` (build.py VARIANTS, -DQTTS_TSTAMP=1): phase timestamps INSIDE the frame step's kernels.
//
// rocprofv3 gives a kernel's dispatch duration; it cannot say how the 5 us of a decode-GEMM launch split into "waiting for the
// first bytes", "MFMA", "cross-wave combine" and "store", nor how long the boundary between two graph nodes is.  In this
// variant wave 0 of the first and of the last workgroup of every instrumented launch reads the 100 MHz constant clock
// (`s_memrealtime`, the same on every CU) at up to 6 points and appends one record to a per-translation-unit log;
// tools/ts_frame.py reads the logs back (`qtts_debug_tslog_<unit>`), orders the records of all units by entry time and prints
// the phase averages per kernel class and the exit -> entry gaps between consecutive launches.
// A drained stamp first waits for every outstanding memory operation of the wave (`s_waitcnt vmcnt(0) lgkmcnt(0)`), i.e. it
// says when the data HAS ARRIVED; that wait also changes the schedule after it, so the variant measures latencies, not the
// product's throughput.  In the product build every macro below expands to nothing.
#pragma once
#ifndef QTTS_TSTAMP
#define QTTS_TSTAMP 0
#endif
#if QTTS_TSTAMP
namespace qtts {
struct TsRec { unsigned long long t[6]; int kind, a, b, blk; };
constexpr unsigned TS_CAP = 1u << 15;
__device__ inline unsigned long long ts_now() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
__device__ inline unsigned long long ts_drained() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
}  // namespace qtts
// one log per translation unit (no relocatable device code in this build: a __device__ variable is private to its unit)
#define QTTS_TS_UNIT(unit)                                                                                                  \
    namespace qtts { __device__ TsRec ts_log_##unit[TS_CAP]; __device__ unsigned ts_cnt_##unit; }                            \
    extern "C" __attribute__((visibility("default"))) int qtts_debug_tslog_##unit(void* out, int max_records) {             \
        unsigned n = 0;                                                                                                      \
        if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(qtts::ts_cnt_##unit), sizeof(n)) != hipSuccess) return -1;                    \
        if (n > qtts::TS_CAP) n = qtts::TS_CAP;                                                                              \
        if ((int)n > max_records) n = (unsigned)max_records;                                                                 \
        if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(qtts::ts_log_##unit), (size_t)n * sizeof(qtts::TsRec)) != hipSuccess)   \
            return -1;                                                                                                       \
        const unsigned zero = 0;                                                                                             \
        if (hipMemcpyToSymbol(HIP_SYMBOL(qtts::ts_cnt_##unit), &zero, sizeof(zero)) != hipSuccess) return -1;                \
        return (int)n;                                                                                                       \
    }
#define QTTS_TS_BEGIN() unsigned long long ts_[6] = {0, 0, 0, 0, 0, 0}; ts_[0] = qtts::ts_now()
#define QTTS_TS(i) ts_[i] = qtts::ts_now()
#define QTTS_TS_DRAINED(i) ts_[i] = qtts::ts_drained()
// wave 0 / lane 0 of the first and the last workgroup (x) append a record
#define QTTS_TS_END(unit, kind_, a_, b_)                                                                                     \
    do {                                                                                                                     \
        if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {                       \
            const unsigned i_ = atomicAdd(&qtts::ts_cnt_##unit, 1u);                                                         \
            if (i_ < qtts::TS_CAP) {                                                                                         \
                qtts::TsRec r_;                                                                                              \
                for (int k_ = 0; k_ < 6; ++k_) r_.t[k_] = ts_[k_];                                                           \
                r_.kind = (kind_); r_.a = (a_); r_.b = (b_); r_.blk = (int)blockIdx.x;                                       \
                qtts::ts_log_##unit[i_] = r_;                                                                                \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
#else
#define QTTS_TS_UNIT(unit)
#define QTTS_TS_BEGIN() do { } while (0)
#define QTTS_TS(i) do { } while (0)
#define QTTS_TS_DRAINED(i) do { } while (0)
#define QTTS_TS_END(unit, kind_, a_, b_) do { } while (0)
#endif
