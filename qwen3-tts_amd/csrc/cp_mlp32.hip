// cp_mlp32.hip -- cp_mlp_kernel's construction (cp_mlp.hip) for batch 9..32: the code predictor's MLP of a layer (RMSNorm -> gate|up GEMM ->
// SwiGLU -> down GEMM -> + residual; modeling_qwen3_tts.py:842-855 inside the decoder layer :961-1012, driven by code_predictor.generate
// :1250-1312) as ONE launch at the batch BASELINE configs 4 and 5 run the frame step at (32 streams / waves of 32 requests).
//
// Round 6.  At batch 17..32 the two decode GEMMs it replaces are bound by the x rows every workgroup pulls (skinny.hip:
// skinny2_ks_kernel's header): gate|up 5.3 us + down 8.4 us + two boundaries of ~2 us per layer on the MI355X (profiles/r06_skinny_ksplit.md).
// Here a workgroup reads the 32 hidden rows ONCE (64 KB) for its 12 + 12 intermediate features, and the down-projection's input reaches it as its
// XCD's slice of the intermediate vector (32 rows x I / 8: 48 KB of granules written by the 32 workgroups that share its L2).
//
// Same phases, same exchange, same summation orders as cp_mlp_kernel -- per row the arithmetic is statement for statement the batch <= 8 kernel's:
//   phase A  workgroup (xcd = b % 8, j = b / 8): ACT gate + ACT up features over K = H for BOTH 16-row tiles (the MFMA's batch columns are all
//            rows now), four waves a quarter of k each, quarters added in wave order, row variances from the same bf16 fragments; wave m of the
//            workgroup finishes tile m: act = silu(gate) * up -> granules {2 x bf16, tag}.
//   phase B  output features [32 j, 32 j + 32) over the XCD's slice: 2 feature tiles x 2 row tiles per wave, quarters added in wave order, the
//            partial sums published as granules {fp32, tag} (four rows per thread).
//   phase C  the workgroups of class 7 add the eight XCD partials in XCD order + residual -> hidden rows (fp32 + bf16 copy).
// Granule buffers have 32 rows per XCD.  All 256 workgroups must be resident (consumers wait for producers of the same launch): the engine's
// per-device account (talker_engine.hip: fused_admit) admits the engine or it keeps the two GEMMs.  bf16 engines only (the exact fp32 mode at
// batch 32 keeps its decode GEMMs; its golden is bit-exact there: tests/test_gpu_parity.py).
#include "common.h"
#include "kernels.h"
#include "tstamp.h"
#include "granule.h"
#include <hip/hip_ext.h>

namespace qtts {

constexpr int MLP32_ROWS = 32;                       // rows per XCD in the granule buffers

bool cp_mlp32_takes(int B, int H, int I) { return B >= 1 && B <= MLP32_ROWS && cp_mlp_takes(1, H, I); }
size_t cp_mlp32_act_bytes(int I) { return (size_t)8 * MLP32_ROWS * (I / 16) * 8; }
size_t cp_mlp32_part_bytes(int H) { return (size_t)8 * MLP32_ROWS * H * 8; }

#define QTTS_CPMLP32_ARGS(P) (P).Wgu, (P).Wd, (P).x16, (P).serial, (P).done_flag, (P).ldx16, (P).slot, (P)
// ACT: intermediate features per workgroup; KQ: k-tiles (of 32) per wave in phase A; KTW: k-tiles of the XCD slice per wave in phase B.
template <int ACT, int KQ, int KTW>
__global__ __launch_bounds__(256) void cp_mlp32_kernel(const void* kWgu, const void* kWd, const unsigned short* kx16, const int* kserial, const int* kdone,
                                                       int kldx16, int kslot, CpMlpParams P) {
    P.Wgu = kWgu; P.Wd = kWd; P.x16 = kx16; P.serial = kserial; P.done_flag = kdone; P.ldx16 = kldx16; P.slot = kslot;
    constexpr int KT = 32, MT = 2, RC = MLP32_ROWS;
    // phase A's k quarters [4 waves][2 row tiles][64 lanes][gate, up] f32x4 + row sums of squares [4][2][16] | phase B's quarters [4][2 feature tiles][2 row tiles][64] f32x4
    constexpr int QA_BYTES = 4 * MT * 64 * 2 * 16 + 4 * MT * 16 * 4, QB_BYTES = 4 * 2 * MT * 64 * 16;
    __shared__ __attribute__((aligned(16))) unsigned char smem[QA_BYTES + QB_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int nktH = P.H / KT, nktI = P.I / KT, slice = P.I >> 3;
    const int spairs = slice >> 1;                        // granules per row of an XCD's slice (a pair of bf16 values each)
    const bool run_a = P.phase == 3 || P.phase == 0, run_b = P.phase == 3 || P.phase == 1, run_c = P.phase == 3 || P.phase == 2;
    const unsigned tag = ((unsigned)*P.serial << 7) | (unsigned)P.slot;
    int rowm[MT];                                         // this lane's row of tile m (rows >= B re-read row 0 and are dropped at use)
#pragma unroll
    for (int m = 0; m < MT; ++m) rowm[m] = m * 16 + li < P.B ? m * 16 + li : 0;
    // ---- 0. phase A's requests: gate / up tiles of this wave's k quarter and the x fragments of both row tiles
    cu32x4 wg[KQ], wu[KQ], gx[MT][KQ], wd[2][KTW];
    {
        const cu32x4* wsrc = reinterpret_cast<const cu32x4*>(P.Wgu) + (((size_t)b * nktH + wave * KQ) * 4 + lq) * (2 * ACT) + (li < ACT ? li : 0);
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            wg[ks] = wsrc[(size_t)ks * 4 * 2 * ACT];
            wu[ks] = wsrc[(size_t)ks * 4 * 2 * ACT + ACT];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const cu32x4* xsrc = reinterpret_cast<const cu32x4*>(P.x16 + (size_t)rowm[m] * P.ldx16 + wave * KQ * 32 + lq * 8);
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks) gx[m][ks] = xsrc[ks * 4];
        }
    }
    // phase B's block of the down operator: requested behind phase A's MFMAs (cp_mlp.hip: it streams while the quarters are combined and the granules travel)
    auto load_wd = [&] {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const cu32x4* dsrc = reinterpret_cast<const cu32x4*>(P.Wd) + ((size_t)(j * 2 + t2) * nktI + xcd * (slice / KT) + wave * KTW) * 64 + lane;
#pragma unroll
            for (int t = 0; t < KTW; ++t) wd[t2][t] = dsrc[t * 64];
        }
    };
    if (!run_a) load_wd();
    const int done = P.done_flag ? *P.done_flag : 0;
    if (done) return;
    f32x4* qa = reinterpret_cast<f32x4*>(smem);
    float* qss = reinterpret_cast<float*>(smem + 4 * MT * 64 * 2 * 16);
    f32x4* qb = reinterpret_cast<f32x4*>(smem + QA_BYTES);
    const WtBuf ag = wt_buf(P.act_gran, (size_t)8 * RC * spairs * 8);
    if (run_a) {
        // ---- A. ACT gate + ACT up features over this wave's k quarter, both row tiles (D[feature 4 q + r][row i]); quarters added in wave order
        f32x4 ag4[MT], au4[MT];
        float ssq[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) { ag4[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; au4[m] = ag4[m]; ssq[m] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            cu32x4 g4 = wg[ks], u4 = wu[ks];
            if (li >= ACT) { g4 = (cu32x4){0u, 0u, 0u, 0u}; u4 = g4; }
            bf16x8 wa, wb2;
            *reinterpret_cast<cu32x4*>(&wa) = g4;
            *reinterpret_cast<cu32x4*>(&wb2) = u4;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                cu32x4 xv4 = gx[m][ks];
                if (m * 16 + li >= P.B) xv4 = (cu32x4){0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(xv4[e] << 16), hi = __uint_as_float(xv4[e] & 0xffff0000u);
                    ssq[m] += lo * lo; ssq[m] += hi * hi;
                }
                bf16x8 xb;
                *reinterpret_cast<cu32x4*>(&xb) = xv4;
                ag4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, ag4[m], 0, 0, 0);
                au4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb2, xb, au4[m], 0, 0, 0);
            }
        }
        load_wd();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ssq[m] += __shfl_xor(ssq[m], 16);
            ssq[m] += __shfl_xor(ssq[m], 32);             // every lane: its row's sum over this wave's k quarter
            qa[((wave * MT + m) * 64 + lane) * 2] = ag4[m];
            qa[((wave * MT + m) * 64 + lane) * 2 + 1] = au4[m];
            if (lq == 0) qss[(wave * MT + m) * 16 + li] = ssq[m];
        }
        __syncthreads();
        if (wave < MT) {                                  // wave m finishes row tile m
            const int m = wave, row = m * 16 + li;
            if (row < P.B && lq * 4 < ACT) {
                f32x4 sg = ((qa[((0 * MT + m) * 64 + lane) * 2] + qa[((1 * MT + m) * 64 + lane) * 2]) + qa[((2 * MT + m) * 64 + lane) * 2]) + qa[((3 * MT + m) * 64 + lane) * 2];
                f32x4 su = ((qa[((0 * MT + m) * 64 + lane) * 2 + 1] + qa[((1 * MT + m) * 64 + lane) * 2 + 1]) + qa[((2 * MT + m) * 64 + lane) * 2 + 1]) +
                           qa[((3 * MT + m) * 64 + lane) * 2 + 1];
                const float ss = ((qss[(0 * MT + m) * 16 + li] + qss[(1 * MT + m) * 16 + li]) + qss[(2 * MT + m) * 16 + li]) + qss[(3 * MT + m) * 16 + li];
                const float rs = rsqrtf(ss / (float)P.H + P.eps);
                float a4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {              // the decode GEMM's SwiGLU epilogue, statement for statement (skinny.hip)
                    const float vg = sg[r] * rs, vu = su[r] * rs;
                    a4[r] = (vg / (1.f + expf(-vg))) * vu;
                }
                const int off = (int)((((size_t)xcd * RC + row) * spairs + ((j * ACT + lq * 4) >> 1)) * 8);
                wt_store16(ag, off, (cu32x4){pack_bf16(a4[0], a4[1]), tag, pack_bf16(a4[2], a4[3]), tag});
            }
        }
    }
    if (!run_b && !run_c) return;
    const WtBuf slab = wt_buf(P.part, (size_t)8 * RC * P.H * 8);
    const int r_t = tid >> 5, f_t = tid & 31;              // phases B' / C: thread = (row r_t + 8 rr, feature of this workgroup's 32)
    float own[4] = {0.f, 0.f, 0.f, 0.f};
    if (run_b) {
        // ---- B. this XCD's slice of the intermediate vector, from the 32 workgroups of this XCD: wait until every granule carries the tag
        int offs[MT][KTW];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < KTW; ++t) offs[m][t] = (int)((((size_t)xcd * RC + rowm[m]) * spairs + (wave * KTW + t) * 16 + lq * 4) * 8);
        cu32x4 cur[MT][KTW][2];                                // (one read set, as in phase C)
        auto load_slice = [&](cu32x4 (&d)[MT][KTW][2]) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < KTW; ++t) { d[m][t][0] = wt_load16(ag, offs[m][t]); d[m][t][1] = wt_load16(ag, offs[m][t] + 16); }
        };
        wt_first_pause(P.first_pause);
        load_slice(cur);
        for (int spins = 0;; ++spins) {
            bool fresh = true;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < KTW; ++t)
#pragma unroll
                    for (int h = 0; h < 2; ++h) fresh = fresh && cur[m][t][h][1] == tag && cur[m][t][h][3] == tag;
            if (fresh) break;
            if (spins > GRANULE_SPIN_LIMIT) {
                if (P.err) __hip_atomic_store(P.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (P.done_latch) __hip_atomic_store(P.done_latch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            asm volatile("" ::: "memory");                 // (the re-read stays in the loop: tests/test_host_logic.py pins it from the ISA)
            wt_first_pause(P.poll_step);
            load_slice(cur);
        }
        f32x4 acc[2][MT];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[t2][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < KTW; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                cu32x4 xv4 = (cu32x4){cur[m][t][0][0], cur[m][t][0][2], cur[m][t][1][0], cur[m][t][1][2]};
                if (m * 16 + li >= P.B) xv4 = (cu32x4){0u, 0u, 0u, 0u};
                bf16x8 xb;
                *reinterpret_cast<cu32x4*>(&xb) = xv4;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    bf16x8 wa;
                    *reinterpret_cast<cu32x4*>(&wa) = wd[t2][t];
                    acc[t2][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[t2][m], 0, 0, 0);
                }
            }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int m = 0; m < MT; ++m) qb[((wave * 2 + t2) * MT + m) * 64 + lane] = acc[t2][m];
        __syncthreads();
        // element (feature f_t, row) of D[feature 4 q + c][row i]: tile (f_t >> 4, row >> 4), lane ((f_t & 15) >> 2) * 16 + (row & 15), component f_t & 3
        const float* qf = reinterpret_cast<const float*>(qb);
        constexpr int WS = 2 * MT * 64 * 4;                 // floats per wave block
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = rr * 8 + r_t;
            const int e = (((((f_t >> 4) * MT + (row >> 4)) * 64) + ((f_t & 15) >> 2) * 16 + (row & 15)) << 2) + (f_t & 3);
            own[rr] = ((qf[e] + qf[WS + e]) + qf[2 * WS + e]) + qf[3 * WS + e];
            if (row < P.B && (xcd != 7 || P.phase != 3)) {
                const int off = (int)((((size_t)xcd * RC + row) * P.H + j * 32 + f_t) * 8);
#ifdef QTTS_HOST_EMU
                *reinterpret_cast<uint2*>(slab.base + off) = uint2{__float_as_uint(own[rr]), tag};
#else
                typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64((cu32x2){__float_as_uint(own[rr]), tag}, slab.r, off, 0, 16);
#endif
            }
        }
    }
    if (!run_c || xcd != 7) return;
    // ---- C. the reducer of output features [32 j, 32 j + 32): the 8 XCD partials in XCD order + the residual, four rows per thread
    {
        const int col = j * 32 + f_t;
        constexpr int NP = 8;
        const int nwait = P.phase == 3 ? NP - 1 : NP;          // (emulator, phase 2 alone: the own partial comes from its slab too)
        float res[4];
        int rowc[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            rowc[rr] = rr * 8 + r_t < P.B ? rr * 8 + r_t : 0;
            res[rr] = P.res[(size_t)rowc[rr] * P.H + col];
        }
        uint2 pa[4][NP];                                       // (one read set: a second one in flight would put the kernel over 224 registers -- two engines per device)
        auto load_slabs = [&](uint2 (&d)[4][NP]) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int x2 = 0; x2 < NP; ++x2) d[rr][x2] = wt_load8(slab, (int)((((size_t)(x2 < nwait ? x2 : 0) * RC + rowc[rr]) * P.H + col) * 8));
        };
        wt_first_pause(P.pause_c);
        load_slabs(pa);
        for (int spins = 0;; ++spins) {
            bool fresh = true;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int x2 = 0; x2 < NP; ++x2) fresh = fresh && (x2 >= nwait || pa[rr][x2].y == tag);
            if (fresh) break;
            if (spins > GRANULE_SPIN_LIMIT) {
                if (P.err) __hip_atomic_store(P.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (P.done_latch) __hip_atomic_store(P.done_latch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            asm volatile("" ::: "memory");                 // (the re-read stays in the loop: tests/test_host_logic.py pins it from the ISA)
            wt_first_pause(P.poll_step);
            load_slabs(pa);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = rr * 8 + r_t;
            if (row >= P.B) continue;
            float s = __uint_as_float(pa[rr][0].x);
#pragma unroll
            for (int x2 = 1; x2 < NP - 1; ++x2) s += __uint_as_float(pa[rr][x2].x);
            s += P.phase == 3 ? own[rr] : __uint_as_float(pa[rr][NP - 1].x);
            s += res[rr];
            P.out[(size_t)row * P.H + col] = s;
            if (P.out16) P.out16[(size_t)row * P.H + col] = f32_to_bf16(s);
        }
    }
}

static thread_local hipEvent_t tl_mlp32_ev_start = nullptr, tl_mlp32_ev_stop = nullptr;
void cp_mlp32_set_launch_events(hipEvent_t start, hipEvent_t stop) { tl_mlp32_ev_start = start; tl_mlp32_ev_stop = stop; }

template <int ACT, int KQ, int KTW>
static void launch_cp_mlp32_t(const CpMlpParams& P, hipStream_t st) {
    const dim3 grid(cp_mlp_grid(P.H));
    auto kern = cp_mlp32_kernel<ACT, KQ, KTW>;
#ifdef QTTS_HOST_EMU
    for (int ph = 0; ph < 3; ++ph) {                        // (the emulator runs workgroups one after the other: the launch runs as its three phases)
        if (P.phase != 3 && P.phase != ph) continue;
        CpMlpParams Q = P;
        Q.phase = ph;
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, QTTS_CPMLP32_ARGS(Q));
    }
#else
    if (tl_mlp32_ev_start) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, st, tl_mlp32_ev_start, tl_mlp32_ev_stop, 0, QTTS_CPMLP32_ARGS(P));
    else hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, QTTS_CPMLP32_ARGS(P));
#endif
}

// (k-tiles of 32: ACT = I / (H / 4), KQ = H / 128, KTW = I / 1024) -- the released width and the emulator tests' 256 / 1024
#define QTTS_CPMLP32_CASES(X) X(12, 8, 3) X(16, 2, 1)

static bool cp_mlp32_shape(int H, int I, int& act, int& kq, int& ktw) {
    if (!cp_mlp_takes(1, H, I)) return false;
    act = I / (H / 4); kq = H / 128; ktw = I / 8 / 32 / 4;
    return true;
}

bool cp_mlp32_instantiated(int H, int I) {
    int act = 0, kq = 0, ktw = 0;
    if (!cp_mlp32_shape(H, I, act, kq, ktw)) return false;
#define QTTS_CPMLP32_X(A, Q, T) if (act == A && kq == Q && ktw == T) return true;
    QTTS_CPMLP32_CASES(QTTS_CPMLP32_X)
#undef QTTS_CPMLP32_X
    return false;
}

void launch_cp_mlp32(const CpMlpParams& P, hipStream_t st) {
    QTTS_REQUIRE(cp_mlp32_takes(P.B, P.H, P.I) && !P.f32, QTTS_ERR_ARG, "cp_mlp32: shape (bf16, batch <= 32, H % 128, I / (H / 4) in {4, 8, 12, 16})");
    QTTS_REQUIRE(P.Wgu && P.Wd && P.x16 && P.res && P.out && P.act_gran && P.part && P.serial, QTTS_ERR_ARG, "cp_mlp32: null operand");
    QTTS_REQUIRE(P.slot >= 0 && P.slot < 128 && P.ldx16 % 8 == 0, QTTS_ERR_ARG, "cp_mlp32: slot must be 0..127, ldx16 % 8");
    int act = 0, kq = 0, ktw = 0;
    cp_mlp32_shape(P.H, P.I, act, kq, ktw);
#define QTTS_CPMLP32_X(A, Q, T) if (act == A && kq == Q && ktw == T) { launch_cp_mlp32_t<A, Q, T>(P, st); QTTS_CHECK_HIP(hipGetLastError()); return; }
    QTTS_CPMLP32_CASES(QTTS_CPMLP32_X)
#undef QTTS_CPMLP32_X
    throw Error(QTTS_ERR_ARG, "cp_mlp32: no instantiation for this (H, I)");
}

int cp_mlp32_blocks_per_cu(int H, int I) {
#ifdef QTTS_HOST_EMU
    if (const char* e = QTTS_ENV("QTTS_HOSTEMU_CPAO_BLOCKS_PER_CU")) return atoi(e);
    return 2;
#else
    int act = 0, kq = 0, ktw = 0;
    if (!cp_mlp32_shape(H, I, act, kq, ktw)) return 0;
    int n = 0;
#define QTTS_CPMLP32_X(A, Q, T) if (act == A && kq == Q && ktw == T) { QTTS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cp_mlp32_kernel<A, Q, T>, 256, 0)); return n; }
    QTTS_CPMLP32_CASES(QTTS_CPMLP32_X)
#undef QTTS_CPMLP32_X
    return 0;
#endif
}

}  // namespace qtts
