// cp_mlp.hip -- the code predictor's MLP of a layer (RMSNorm -> gate|up GEMM -> SwiGLU -> down GEMM -> + residual) as ONE launch.
// Replaces, for the passes of `code_predictor.generate` at batch <= 8 (modeling_qwen3_tts.py:1250-1312 driving the decoder layer's MLP,
// :880-958 / Qwen3TTSTalkerTextMLP :820-832), two launches of the decode GEMM (skinny8_kernel ACT_SWIGLU8 + the down-projection) and
// the dependent-launch boundary between them: 75 of each per frame.
//
// Round 5 ("measure one XCD-local MLP fusion").  The construction is cp_attn_o_kernel's (attention.hip): values
// cross from workgroup to workgroup INSIDE the launch as tagged 8-byte granules (granule.h), every edge narrow, every summation order
// fixed.  What is new is the shape of the exchange.  The down-projection needs ALL of the intermediate vector (I = 3072) for every
// output feature -- an all-to-all between the workgroups that produced it.  Here that all-to-all is cut by XCD:
//   workgroup b = (xcd = b % 8, j = b / 8): consecutive workgroups of a launch go to consecutive XCDs, so the 32 workgroups with the
//   same `xcd` share one L2.
//   phase A  gate|up: XCD x owns the slice [x I/8, (x + 1) I/8) of the intermediate vector; its workgroup j computes ACT = I / grid
//            features of it (12 at 1024 / 3072): two MFMA tiles (ACT gate rows, ACT up rows; rows >= ACT are padding) over K = H, the
//            four waves a quarter of k each, quarters added in wave order, rsqrt(mean x^2 + eps) from the same bf16 x fragments (the
//            RMSNorm weight is folded into the operator), act = silu(gate) * up rounded to bf16 as between the two separate launches,
//            published as granules {2 x bf16, tag} -- 48 KB of weights per workgroup, requested at kernel entry.
//   phase B  partial down-projection: workgroup (x, j) owns output features [32 j, 32 j + 32) and the k range of ITS XCD's slice: it
//            reads the slice (8 rows x I / 8 values = 12 KB of granules, written by the 32 workgroups of its own XCD) and multiplies it
//            with its [32 x I / 8] block of the operator (24 KB, requested at kernel entry: it streams while phase A runs and the
//            granules travel).  The four waves take a quarter of the slice each; quarters added in wave order.
//   phase C  the 8 XCD partials of every (row, feature) are added in XCD order by the workgroup of XCD 7 (its own partial never leaves
//            the workgroup) + the residual -> hidden state (fp32 + bf16 copy).  Only these 8 x H partial rows cross XCDs.
// Nobody waits before it has produced: phase A depends on nothing inside the launch, phase B on phase A of the same XCD, phase C on
// phase B -- no circular wait, provided all workgroups are resident (the engine's admission rule, talker_engine.hip: fused_admit).
// A consumer that loses its producers gives up and latches the generation's stop flag (the cold block of its polling loop: nothing
// else of the loop may depend on it -- attention.hip: cpao_give_up): what the launch still writes is never consumed.
// bf16 engines, batch <= 8, H % 128 == 0, (I / (H / 4)) in {4, 8, 12, 16}; everything else keeps the two launches.  (24 KB of the down
// operator: requested behind phase A's MFMAs since the A/B of profiles/r05_cp_mlp.md.)
// fp32 engines (the exact parity mode), on request (QTTS_CP_MLP_F32=1): the F32 instantiation of the same source -- fp32 operators, rows and
// intermediate vector -- through which the reference's fp32 goldens run bit-exact on the MI355X (tests/test_gpu_parity.py:
// test_fused_launches_fp32_instantiations_bit_exact_vs_reference_golden); slower there than the split-K plan, hence not the default.
#include "common.h"
#include "kernels.h"
#include "tstamp.h"
#include "granule.h"
#include <hip/hip_ext.h>

QTTS_TS_UNIT(cpmlp)

namespace qtts {

// Packed gate|up operator of the fused launch: [workgroup b = xcd + 8 j][H / 32 k-tiles][4 k-slices][2 ACT rows: ACT gate | ACT up][8 bf16],
// RMSNorm weight g folded in.  Workgroup (xcd, j) owns intermediate features xcd * (I / 8) + j * ACT + r.
// fp32 engines (the exact parity mode): the same with k-tiles of 16 and 4 floats per 16-byte unit.
size_t cp_mlp_gu_bytes(int H, int I, bool bf16) { return (size_t)2 * I * H * (bf16 ? 2 : 4); }
void pack_cp_mlp_gu(const float* Wg, const float* Wu, const float* g, int H, int I, bool bf16, void* out_host) {
    const int J = H / 32, ACT = I / (8 * J), KT = bf16 ? 32 : 16, E = bf16 ? 8 : 4, nkt = H / KT;
    parallel_for(8 * J, [&](int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; ++b) {
            const int xcd = (int)(b & 7), j = (int)(b >> 3);
            for (int kt = 0; kt < nkt; ++kt)
                for (int q = 0; q < 4; ++q)
                    for (int r2 = 0; r2 < 2 * ACT; ++r2) {
                        const int f = xcd * (I / 8) + j * ACT + (r2 % ACT);
                        const float* src = (r2 < ACT ? Wg : Wu) + (size_t)f * H + kt * KT + q * E;
                        const size_t unit = (((size_t)b * nkt + kt) * 4 + q) * (2 * ACT) + r2;
                        for (int e = 0; e < E; ++e) {
                            const float v = g ? src[e] * g[kt * KT + q * E + e] : src[e];
                            if (bf16) reinterpret_cast<bf16_t*>(out_host)[unit * 8 + e] = f32_to_bf16(v);
                            else reinterpret_cast<float*>(out_host)[unit * 4 + e] = v;
                        }
                    }
        }
    });
}

bool cp_mlp_takes(int B, int H, int I) {
    if (B < 1 || B > 8 || H % 128 != 0 || H < 128 || I % (H / 4) != 0) return false;
    const int act = I / (H / 4);                       // = I / (8 J), J = H / 32
    if (act != 4 && act != 8 && act != 12 && act != 16) return false;
    const int ktw = I / 8 / 32 / 4;                    // k-tiles of an XCD's slice per wave
    return ktw >= 1 && ktw <= 4 && (I / 8) % 128 == 0 && H / 128 <= 8;
}
int cp_mlp_grid(int H) { return 8 * (H / 32); }

#if QTTS_TSTAMP
#define QTTS_TS_CPMLP(tail_)                                                                                             \
    if (threadIdx.x == 0 && ((tail_) || blockIdx.x % 9 == 4)) {                                                          \
        const unsigned i_ = atomicAdd(&qtts::ts_cnt_cpmlp, 1u);                                                          \
        if (i_ < qtts::TS_CAP) {                                                                                         \
            qtts::TsRec r_;                                                                                              \
            for (int k_ = 0; k_ < 6; ++k_) r_.t[k_] = ts_[k_];                                                           \
            r_.kind = 5; r_.a = P.slot; r_.b = 0; r_.blk = (int)blockIdx.x | ((tail_) << 16);                            \
            qtts::ts_log_cpmlp[i_] = r_;                                                                                 \
        }                                                                                                                \
    }
#else
#define QTTS_TS_CPMLP(tail_)
#endif
#define QTTS_CPMLP_ARGS(P) (P).Wgu, (P).Wd, (P).x16, (P).serial, (P).done_flag, (P).ldx16, (P).slot, (P)
// ACT: intermediate features per workgroup; KQ: k-tiles per wave in phase A; KTW: k-tiles of the XCD slice per wave in phase B.
// F32 (the exact parity mode): fp32 operators and fp32 x rows (`x16` then points to floats), k-tiles of 16 consumed by four
// v_mfma_f32_16x16x4_f32 each, the intermediate vector travels as fp32 (one value per granule) -- as between the fp32 engines' separate launches.
template <bool F32, int ACT, int KQ, int KTW>
__global__ __launch_bounds__(256) void cp_mlp_kernel(const void* kWgu, const void* kWd, const unsigned short* kx16, const int* kserial, const int* kdone,
                                                     int kldx16, int kslot, CpMlpParams P) {
    P.Wgu = kWgu; P.Wd = kWd; P.x16 = kx16; P.serial = kserial; P.done_flag = kdone; P.ldx16 = kldx16; P.slot = kslot;
    constexpr int KT = F32 ? 16 : 32;                   // k per tile
    // ONE LDS object: phase A's k quarters [4 waves][64 lanes][gate, up] f32x4 + row sums of squares [4][16] | phase B's quarters [4][2 tiles][64] f32x4
    constexpr int QA_BYTES = 4 * 64 * 2 * 16 + 4 * 16 * 4, QB_BYTES = 4 * 2 * 64 * 16;
    __shared__ __attribute__((aligned(16))) unsigned char smem[QA_BYTES + QB_BYTES];
    QTTS_TS_BEGIN();                       // (tstamp build: 1 = phase A's operands arrived, 2 = act published, 3 = slice read, 4 = partial published, 5 = reduced)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int nktH = P.H / KT, nktI = P.I / KT, slice = P.I >> 3;
    const int spairs = F32 ? slice : slice >> 1;      // granules per row of an XCD's slice (bf16: a pair of values each; fp32: one)
    const bool run_a = P.phase == 3 || P.phase == 0, run_b = P.phase == 3 || P.phase == 1, run_c = P.phase == 3 || P.phase == 2;
    const unsigned tag = ((unsigned)*P.serial << 7) | (unsigned)P.slot;
    // ---- 0. every weight request of the launch: phase A's gate / up tiles of this wave's k quarter, then phase B's block of the down operator
    cu32x4 wg[KQ], wu[KQ], gx[KQ], wd[2][KTW];
    {
        const cu32x4* wsrc = reinterpret_cast<const cu32x4*>(P.Wgu) + (((size_t)b * nktH + wave * KQ) * 4 + lq) * (2 * ACT) + (li < ACT ? li : 0);
        const cu32x4* xsrc = F32 ? reinterpret_cast<const cu32x4*>(reinterpret_cast<const float*>(P.x16) + (size_t)(li < P.B ? li : 0) * P.ldx16 + wave * KQ * 16 + lq * 4)
                                 : reinterpret_cast<const cu32x4*>(P.x16 + (size_t)(li < P.B ? li : 0) * P.ldx16 + wave * KQ * 32 + lq * 8);
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            wg[ks] = wsrc[(size_t)ks * 4 * 2 * ACT];
            wu[ks] = wsrc[(size_t)ks * 4 * 2 * ACT + ACT];
            gx[ks] = xsrc[ks * 4];
        }
    }
    // phase B's block of the down operator: requested BEHIND phase A's operands are consumed (its 24 KB would share the workgroup's memory
    // pipe with the 48 KB phase A waits for; it streams while the quarters are combined and the granules travel) (at entry -- round 5's first version -- the frame was 1.8 % slower: profiles/r05_cp_mlp.md; switch retired in round 6)
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
    float* qss = reinterpret_cast<float*>(smem + 4 * 64 * 2 * 16);
    f32x4* qb = reinterpret_cast<f32x4*>(smem + QA_BYTES);
    const WtBuf ag = wt_buf(P.act_gran, (size_t)8 * 8 * spairs * 8);
    if (run_a) {
        // ---- A. ACT gate + ACT up features over this wave's k quarter (D[feature 4 q + r][sequence i]); quarters added in wave order
        f32x4 ag4 = (f32x4){0.f, 0.f, 0.f, 0.f}, au4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        float ssq = 0.f;
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            cu32x4 xv4 = gx[ks], g4 = wg[ks], u4 = wu[ks];
            if (li >= P.B) xv4 = (cu32x4){0u, 0u, 0u, 0u};
            if (li >= ACT) { g4 = (cu32x4){0u, 0u, 0u, 0u}; u4 = g4; }
            if constexpr (F32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xe = __uint_as_float(xv4[e]);
                    ssq += xe * xe;
                    ag4 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(g4[e]), xe, ag4, 0, 0, 0);
                    au4 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(u4[e]), xe, au4, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(xv4[e] << 16), hi = __uint_as_float(xv4[e] & 0xffff0000u);
                    ssq += lo * lo; ssq += hi * hi;
                }
                bf16x8 wa, wb2, xb;
                *reinterpret_cast<cu32x4*>(&wa) = g4;
                *reinterpret_cast<cu32x4*>(&wb2) = u4;
                *reinterpret_cast<cu32x4*>(&xb) = xv4;
                ag4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, ag4, 0, 0, 0);
                au4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb2, xb, au4, 0, 0, 0);
            }
        }
        load_wd();
        ssq += __shfl_xor(ssq, 16);
        ssq += __shfl_xor(ssq, 32);                      // every lane: its row's sum over this wave's k quarter
        qa[(wave * 64 + lane) * 2] = ag4;
        qa[(wave * 64 + lane) * 2 + 1] = au4;
        if (lq == 0) qss[wave * 16 + li] = ssq;
        QTTS_TS(1);
        __syncthreads();
        if (wave == 0 && li < P.B && lq * 4 < ACT) {
            f32x4 sg = ((qa[lane * 2] + qa[(64 + lane) * 2]) + qa[(128 + lane) * 2]) + qa[(192 + lane) * 2];
            f32x4 su = ((qa[lane * 2 + 1] + qa[(64 + lane) * 2 + 1]) + qa[(128 + lane) * 2 + 1]) + qa[(192 + lane) * 2 + 1];
            const float ss = ((qss[li] + qss[16 + li]) + qss[32 + li]) + qss[48 + li];
            const float rs = rsqrtf(ss / (float)P.H + P.eps);
            float a4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                  // the decode GEMM's SwiGLU epilogue, statement for statement (skinny.hip: ACT_SWIGLU8)
                const float vg = sg[r] * rs, vu = su[r] * rs;
                a4[r] = (vg / (1.f + expf(-vg))) * vu;
            }
            if constexpr (F32) {
                const int off = (int)((((size_t)xcd * 8 + li) * spairs + j * ACT + lq * 4) * 8);
                wt_store16(ag, off, (cu32x4){__float_as_uint(a4[0]), tag, __float_as_uint(a4[1]), tag});
                wt_store16(ag, off + 16, (cu32x4){__float_as_uint(a4[2]), tag, __float_as_uint(a4[3]), tag});
            } else {
                const int off = (int)((((size_t)xcd * 8 + li) * spairs + ((j * ACT + lq * 4) >> 1)) * 8);
                wt_store16(ag, off, (cu32x4){pack_bf16(a4[0], a4[1]), tag, pack_bf16(a4[2], a4[3]), tag});
            }
        }
        QTTS_TS(2);
    }
    if (!run_b && !run_c) return;
    const WtBuf slab = wt_buf(P.part, (size_t)8 * 8 * P.H * 8);
    const int r_t = tid >> 5, f_t = tid & 31;              // phases B' / C: thread = (row, feature of this workgroup's 32)
    float own = 0.f;
    if (run_b) {
        // ---- B. this XCD's slice of the intermediate vector, from the 32 workgroups of this XCD: wait until every granule carries the tag
        const int row = li < P.B ? li : 0;
        int offs[KTW];
#pragma unroll
        for (int t = 0; t < KTW; ++t) offs[t] = (int)((((size_t)xcd * 8 + row) * spairs + (wave * KTW + t) * 16 + lq * 4) * 8);       // (bf16: 16 pairs = 32 k per tile; fp32: 16 values)
        cu32x4 cur[KTW][2], nxt[KTW][2];
        auto load_slice = [&](cu32x4 (&d)[KTW][2]) {
#pragma unroll
            for (int t = 0; t < KTW; ++t) { d[t][0] = wt_load16(ag, offs[t]); d[t][1] = wt_load16(ag, offs[t] + 16); }
        };
        wt_first_pause(P.first_pause);
        load_slice(cur);
        wt_first_pause(P.poll_step);
        load_slice(nxt);
        for (int spins = 0;; ++spins) {
            bool fresh = true;
#pragma unroll
            for (int t = 0; t < KTW; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) fresh = fresh && cur[t][h][1] == tag && cur[t][h][3] == tag;
            if (fresh) break;
            if (spins > GRANULE_SPIN_LIMIT) {
                if (P.err) __hip_atomic_store(P.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (P.done_latch) __hip_atomic_store(P.done_latch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
#pragma unroll
            for (int t = 0; t < KTW; ++t) { cur[t][0] = nxt[t][0]; cur[t][1] = nxt[t][1]; }
            wt_first_pause(P.poll_step);
            load_slice(nxt);
        }
        QTTS_TS(3);
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
        for (int t = 0; t < KTW; ++t) {
            cu32x4 xv4 = (cu32x4){cur[t][0][0], cur[t][0][2], cur[t][1][0], cur[t][1][2]};
            if (li >= P.B) xv4 = (cu32x4){0u, 0u, 0u, 0u};
            if constexpr (F32) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[t2] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wd[t2][t][e]), __uint_as_float(xv4[e]), acc[t2], 0, 0, 0);
            } else {
                bf16x8 xb;
                *reinterpret_cast<cu32x4*>(&xb) = xv4;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    bf16x8 wa;
                    *reinterpret_cast<cu32x4*>(&wa) = wd[t2][t];
                    acc[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[t2], 0, 0, 0);
                }
            }
        }
        qb[(wave * 2 + 0) * 64 + lane] = acc[0];
        qb[(wave * 2 + 1) * 64 + lane] = acc[1];
        __syncthreads();
        // element (feature f_t of this workgroup's 32, row r_t) of D[feature 4 q + c][sequence i]: tile f_t >> 4, lane ((f_t & 15) >> 2) * 16 + r_t, component f_t & 3
        const float* qf = reinterpret_cast<const float*>(qb);
        const int e = (((f_t >> 4) * 64 + ((f_t & 15) >> 2) * 16 + r_t) << 2) + (f_t & 3);
        own = ((qf[e] + qf[2 * 64 * 4 + e]) + qf[4 * 64 * 4 + e]) + qf[6 * 64 * 4 + e];
        if (r_t < P.B && (xcd != 7 || P.phase != 3)) {
            const int off = (int)((((size_t)xcd * 8 + r_t) * P.H + j * 32 + f_t) * 8);
            // (an 8-byte write-through store through the same descriptor as the 16-byte ones)
#ifdef QTTS_HOST_EMU
            *reinterpret_cast<uint2*>(slab.base + off) = uint2{__float_as_uint(own), tag};
#else
            typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64((cu32x2){__float_as_uint(own), tag}, slab.r, off, 0, 16);
#endif
        }
        QTTS_TS(4);
    }
    if (!run_c || xcd != 7) { QTTS_TS_CPMLP(0) return; }
    // ---- C. the reducer of output features [32 j, 32 j + 32): the 8 XCD partials in XCD order + the residual
    if (r_t < P.B) {
        const int col = j * 32 + f_t;
        const float res = P.res[(size_t)r_t * P.H + col];
        constexpr int NP = 8;
        const int nwait = P.phase == 3 ? NP - 1 : NP;          // (emulator, phase 2 alone: the own partial comes from its slab too)
        uint2 pa[NP], pn[NP];
        auto load_slabs = [&](uint2 (&d)[NP]) {
#pragma unroll
            for (int x2 = 0; x2 < NP; ++x2) d[x2] = wt_load8(slab, (int)((((size_t)(x2 < nwait ? x2 : 0) * 8 + r_t) * P.H + col) * 8));
        };
        wt_first_pause(P.pause_c);
        load_slabs(pa);
        wt_first_pause(P.poll_step);
        load_slabs(pn);
        for (int spins = 0;; ++spins) {
            bool fresh = true;
#pragma unroll
            for (int x2 = 0; x2 < NP; ++x2) fresh = fresh && (x2 >= nwait || pa[x2].y == tag);
            if (fresh) break;
            if (spins > GRANULE_SPIN_LIMIT) {
                if (P.err) __hip_atomic_store(P.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (P.done_latch) __hip_atomic_store(P.done_latch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
#pragma unroll
            for (int x2 = 0; x2 < NP; ++x2) pa[x2] = pn[x2];
            wt_first_pause(P.poll_step);
            load_slabs(pn);
        }
        float s = __uint_as_float(pa[0].x);
#pragma unroll
        for (int x2 = 1; x2 < NP - 1; ++x2) s += __uint_as_float(pa[x2].x);
        s += P.phase == 3 ? own : __uint_as_float(pa[NP - 1].x);
        s += res;
        P.out[(size_t)r_t * P.H + col] = s;
        if (P.out16) P.out16[(size_t)r_t * P.H + col] = f32_to_bf16(s);
    }
    QTTS_TS_DRAINED(5);
    QTTS_TS_CPMLP(1)
}

static thread_local hipEvent_t tl_mlp_ev_start = nullptr, tl_mlp_ev_stop = nullptr;
void cp_mlp_set_launch_events(hipEvent_t start, hipEvent_t stop) { tl_mlp_ev_start = start; tl_mlp_ev_stop = stop; }

template <bool F32, int ACT, int KQ, int KTW>
static void launch_cp_mlp_t(const CpMlpParams& P, hipStream_t st) {
    const dim3 grid(cp_mlp_grid(P.H));
    auto kern = cp_mlp_kernel<F32, ACT, KQ, KTW>;
#ifdef QTTS_HOST_EMU
    // The emulator runs the workgroups of a launch one after the other: the launch runs as its three phases (the same code, the same tag).
    for (int ph = 0; ph < 3; ++ph) {
        if (P.phase != 3 && P.phase != ph) continue;       // (a single phase given: that phase alone -- the stale-granule test)
        CpMlpParams Q = P;
        Q.phase = ph;
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, QTTS_CPMLP_ARGS(Q));
    }
#else
    if (tl_mlp_ev_start) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, st, tl_mlp_ev_start, tl_mlp_ev_stop, 0, QTTS_CPMLP_ARGS(P));
    else hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, QTTS_CPMLP_ARGS(P));
#endif
}

// (bf16: k-tiles of 32, KQ = H / 128, KTW = I / 1024; fp32: k-tiles of 16, KQ = H / 64, KTW = I / 512)
#define QTTS_CPMLP_CASES(X) X(false, 12, 8, 3) X(false, 16, 2, 1) X(false, 16, 8, 4) X(false, 8, 8, 2) X(false, 4, 8, 1) X(false, 16, 4, 2) X(false, 8, 4, 1) \
                            X(true, 12, 16, 6) X(true, 16, 4, 2)

static bool cp_mlp_shape(int H, int I, bool f32, int& act, int& kq, int& ktw) {
    if (!cp_mlp_takes(1, H, I)) return false;
    act = I / (H / 4); kq = H / (f32 ? 64 : 128); ktw = I / 8 / (f32 ? 16 : 32) / 4;
    return true;
}

void launch_cp_mlp(const CpMlpParams& P, hipStream_t st) {
    QTTS_REQUIRE(cp_mlp_takes(P.B, P.H, P.I), QTTS_ERR_ARG, "cp_mlp: shape (batch <= 8, H % 128, I / (H / 4) in {4, 8, 12, 16})");
    QTTS_REQUIRE(P.Wgu && P.Wd && P.x16 && P.res && P.out && P.act_gran && P.part && P.serial, QTTS_ERR_ARG, "cp_mlp: null operand");
    QTTS_REQUIRE(P.slot >= 0 && P.slot < 128 && P.ldx16 % 8 == 0 && (!P.f32 || !P.out16), QTTS_ERR_ARG, "cp_mlp: slot must be 0..127, ldx16 % 8, no bf16 copy in fp32 mode");
    int act = 0, kq = 0, ktw = 0;
    cp_mlp_shape(P.H, P.I, P.f32 != 0, act, kq, ktw);
    const bool f32 = P.f32 != 0;
#define QTTS_CPMLP_X(F, A, Q, T) if (f32 == F && act == A && kq == Q && ktw == T) { launch_cp_mlp_t<F, A, Q, T>(P, st); QTTS_CHECK_HIP(hipGetLastError()); return; }
    QTTS_CPMLP_CASES(QTTS_CPMLP_X)
#undef QTTS_CPMLP_X
    throw Error(QTTS_ERR_ARG, "cp_mlp: no instantiation for this (H, I, dtype)");
}

bool cp_mlp_instantiated(int H, int I, bool bf16) {
    int act = 0, kq = 0, ktw = 0;
    if (!cp_mlp_shape(H, I, !bf16, act, kq, ktw)) return false;
    const bool f32 = !bf16;
#define QTTS_CPMLP_X(F, A, Q, T) if (f32 == F && act == A && kq == Q && ktw == T) return true;
    QTTS_CPMLP_CASES(QTTS_CPMLP_X)
#undef QTTS_CPMLP_X
    return false;
}

int cp_mlp_blocks_per_cu(int H, int I, bool bf16) {
#ifdef QTTS_HOST_EMU
    if (const char* e = QTTS_ENV("QTTS_HOSTEMU_CPAO_BLOCKS_PER_CU")) return atoi(e);
    return 2;
#else
    int act = 0, kq = 0, ktw = 0;
    if (!cp_mlp_shape(H, I, !bf16, act, kq, ktw)) return 0;
    const bool f32 = !bf16;
    int n = 0;
#define QTTS_CPMLP_X(F, A, Q, T) if (f32 == F && act == A && kq == Q && ktw == T) { QTTS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cp_mlp_kernel<F, A, Q, T>, 256, 0)); return n; }
    QTTS_CPMLP_CASES(QTTS_CPMLP_X)
#undef QTTS_CPMLP_X
    return 0;
#endif
}

}  // namespace qtts
