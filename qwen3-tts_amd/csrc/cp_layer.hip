// cp_layer.hip -- a WHOLE decoder layer of the code predictor (passes >= 1, batch <= 8) as ONE launch:
//   [q|k|v GEMM] -> q/k RMSNorm + RoPE + attention over <= 16 keys -> o-projection (+ residual)
//   -> RMSNorm -> gate|up GEMM -> SwiGLU -> down GEMM (+ residual)
// i.e. Qwen3TTSDecoderLayer.forward (modeling_qwen3_tts.py:961-1012) with its attention (:885-958) and MLP (:842-855) as driven by
// `code_predictor.generate` (:1250-1312).  Round 6: rounds 4-5 ran the layer as TWO launches -- cp_attn_o_kernel
// (attention.hip) and cp_mlp_kernel (cp_mlp.hip) -- each of which starts cold: in cp_mlp's own timeline 3.9 of its 8.6 us pass before
// the first MFMA (launch latency + the first bytes of 48 KB of gate|up per workgroup).  Here the two kernels' stages run in one
// launch of 256 workgroups x 4 waves, statement for statement the same arithmetic in the same summation orders, and
//   * the launch boundary between them becomes one more tagged-granule hand-off (granule.h): the reducers of the o-projection
//     publish the hidden rows as {2 x bf16, tag} granules and every workgroup's waves read the k quarter they own (8 KB per wave and
//     round);
//   * the workgroup's 48 KB of the gate|up operator are requested AT KERNEL ENTRY by LDS-DMA (`global_load_lds`, no staging
//     registers: the attention stage keeps its ~180) and lie in LDS long before the hidden rows arrive -- phase A starts when x
//     arrives, not one cold start later; the 24 KB of the down operator are requested while the workgroup waits for the hidden rows.
// Workgroup b plays (row pair rq, kv head g, 128-feature chunk c) = cp_attn_o's role in the attention stage and (xcd = b % 8,
// j = b / 8) = cp_mlp's role in the MLP stage.  The reducer of the MLP's partial sums is the workgroup that reduced the
// o-projection for the same (row pair, chunk): the residual (the hidden rows after attention) never leaves its registers.
//   stage 0   q|k|v strip (QKV: layers >= 1; layer 0's row comes from the table)      -> strip granules      [all-to-few: 6 strips per wave]
//   stage 1   attention + partial o-projection over the 256 k of kv head g               -> partial granules    [8 -> 1 per (row pair, chunk)]
//   stage 2   reducers (g == 7): sum in kv-head order + residual = hidden x1             -> hidden granules     [32 -> all: the former launch boundary]
//   stage 3   phase A: ACT gate + ACT up features of the XCD's slice, SwiGLU             -> act granules        [XCD-local, 32 -> 32]
//   stage 4   phase B: 32 output features over the XCD's slice of the intermediate vector -> partial granules    [8 -> 1 per (row pair, chunk)]
//   stage 5   reducers: sum in XCD order + x1 = hidden x2 -> global rows (fp32 + bf16 copy) for the next launch
// Nobody waits before it has produced what others wait for at an EARLIER stage, and every wait is for an earlier stage: no circular
// wait, provided all 256 workgroups are resident (talker_engine.hip: fused_admit accounts registers AND LDS for this kernel).
// One launch tag for all five buffers: each is written once per launch.  A consumer that loses its producers gives up and latches the
// generation's stop flag (cold block only; attention.hip: cpao_give_up explains why nothing else of a polling loop may depend on it).
// bf16 engines, H = 1024 (the q|k|v front's strip count = the grid), batch <= 8, contiguous page table; fp32 engines on request
// (F32: the exact parity mode's instantiation -- fp32 operators in registers, fp32 rows and granules -- the construction's bit-exact leg).
#include "common.h"
#include "kernels.h"
#include "tstamp.h"
#include "granule.h"
#include "attn_helpers.h"
#include <hip/hip_ext.h>

QTTS_TS_UNIT(cplayer)

namespace qtts {

__device__ __forceinline__ void cl_dma16(const void* src, void* lds_dst) {        // 64 lanes x 16 B -> 1 KiB of LDS, lane-linear
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// This wave's LDS-DMA has landed when ITS vmcnt says so (each wave reads back only what it requested).  Vector loads complete in issue
// order, so "all but the NEWER newest requests" is enough: the polling loop that runs between the DMA and its use keeps a second round of
// reads in flight, and vmcnt(0) would wait a memory round trip for reads whose values nobody uses (first timeline of this kernel: 1.1 us).
template <int NEWER>
__device__ __forceinline__ void cl_dma_wait() {
#ifndef QTTS_HOST_EMU
    static_assert(NEWER >= 0 && NEWER < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER) : "memory");
#else
    __builtin_amdgcn_wave_barrier();                 // (emulator: lanes are fibers -- every lane of the wave has issued its copies before any lane reads them back)
#endif
}
__device__ __forceinline__ void cl_give_up(const CpLayerParams& P) {
    if (P.ao.err) __hip_atomic_store(P.ao.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (P.ao.done_latch) __hip_atomic_store(P.ao.done_latch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#if QTTS_TSTAMP
#define QTTS_TS_CPLAYER(tail_)                                                                                           \
    if (threadIdx.x == 0 && ((tail_) || blockIdx.x % 9 == 4)) {                                                          \
        const unsigned i_ = atomicAdd(&qtts::ts_cnt_cplayer, 1u);                                                        \
        if (i_ < qtts::TS_CAP) {                                                                                         \
            qtts::TsRec r_;                                                                                              \
            for (int k_ = 0; k_ < 6; ++k_) r_.t[k_] = ts_[k_];                                                           \
            r_.kind = 6; r_.a = P.ao.slot; r_.b = 0; r_.blk = (int)blockIdx.x | ((tail_) << 16);                         \
            qtts::ts_log_cplayer[i_] = r_;                                                                               \
        }                                                                                                                \
    }
#else
#define QTTS_TS_CPLAYER(tail_)
#endif

// LDS of one workgroup (dynamic: > 64 KB with the DMA region): [gate|up block (DMA; bf16 only)] | attention stage / MLP stage
template <bool QKV, bool F32, int ACT, int KQ>
struct ClLds {
    static constexpr int BSTR = 264;
    static constexpr int GU = F32 ? 0 : KQ * 4 * 4 * 2 * ACT * 16;         // [4 waves x KQ k-tiles][4 k-slices][ACT gate | ACT up rows][16 B]
    static constexpr int WS = 4 * 1536, BT = 2 * BSTR * (F32 ? 4 : 2), OWN = 2 * 128 * 4, QP = QKV ? 4 * 64 * 16 + 4 * 16 * 4 : 0;
    static constexpr int ATT = WS + BT + OWN + QP;
    static constexpr int QA = 4 * 64 * 2 * 16 + 4 * 16 * 4, QB = 4 * 2 * 64 * 16;
    static constexpr int MLP = QA + QB;
    // the MLP stage's quarters lie OVER the attention stage's region (dead by then; one workgroup barrier in between): 64.6 KB per workgroup at the
    // released dims, two workgroups per compute unit with a fifth of the LDS to spare
    static constexpr int TOTAL = GU + (ATT > MLP ? ATT : MLP);
};

#define QTTS_CPLAYER_ARGS(P) ((P).ao.Wqkv ? (P).ao.Wqkv : static_cast<const void*>((P).ao.a.qkv)), (P).mlp.Wgu, (P).ao.x16, (P).ao.serial, (P).ao.a.done_flag, (P).ao.a.B, (P).ao.slot, (P)
// phase: CL_ALL on the device.  The host emulator runs the workgroups of a launch one after the other, so there the launch runs as its
// stages (the same code, the same tag): 0 strips | 1 attention + o-projection + reduce | 2 phase A | 3 phase B | 4 the MLP's reduce.
constexpr int CL_ALL = 8;
template <bool QKV, bool F32, int ACT, int KQ, int KTW>
__global__ __launch_bounds__(256) void cp_layer_kernel(const void* k0, const void* kWgu, const unsigned short* kx16, const int* kserial, const int* kdone, int kB,
                                                       int kslot, CpLayerParams P) {
    if constexpr (QKV) P.ao.Wqkv = k0; else P.ao.a.qkv = static_cast<const float*>(k0);
    P.mlp.Wgu = kWgu; P.ao.x16 = kx16; P.ao.serial = kserial; P.ao.a.done_flag = kdone; P.ao.a.B = kB; P.ao.slot = kslot;
    typedef ClLds<QKV, F32, ACT, KQ> L;
    constexpr int HD = 128, MAXK = 16, KW = F32 ? 8 : 4, NKV = 8, BSTR = L::BSTR;
    constexpr int KT = F32 ? 16 : 32;                   // k per tile of the packed operators
    constexpr int NKS = F32 ? 16 : 8;                   // k-tiles per wave: a quarter of K = 1024 in the front, the 256 k of a kv head in the o-projection
    typedef typename CpaoKv<F32>::type KVT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cl[];
    unsigned char* gu_lds = smem_cl;
    unsigned char* att_lds = smem_cl + L::GU;
    unsigned char* mlp_lds = att_lds;                   // (aliased: see ClLds)
    const AttnDecodeParams& p = P.ao.a;
    const int H = P.ao.H, I = P.mlp.I, B = p.B;
    const int nchunk = H >> 7;
    const int b_ = blockIdx.x;
    const int rq = b_ / (NKV * nchunk), gc = b_ - rq * (NKV * nchunk);
    const int g = gc / nchunk, c = gc - g * nchunk;     // attention stage: (row pair, kv head, chunk)
    const int xcd = b_ & 7, j = b_ >> 3;                // MLP stage: (XCD class, slice index)
    QTTS_TS_BEGIN();                       // (tstamp build: 1 = partial o-projection stored, 2 = hidden rows read, 3 = act published, 4 = partial published, 5 = reduced)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = wave & 1, hh = wave >> 1;
    const int kk = lane >> 2, qq = lane & 3;
    const int li = lane & 15, lq = lane >> 4;
    const int S0 = p.len_static, S1 = S0 + 1;
    const int row_w = rq * 2 + rr;
    const bool have = row_w < B;
    const int b = have ? row_w : 0;
    const bool att_wg = rq * 2 < B;                    // this row pair holds a sequence: the workgroup attends, projects and (g == 7) reduces
    const bool reducer = g == NKV - 1 && att_wg;
    const int ph = P.phase;
    const bool run0 = QKV && (ph == CL_ALL || ph == 0), run1 = ph == CL_ALL || ph == 1, run2 = ph == CL_ALL || ph == 2, run3 = ph == CL_ALL || ph == 3,
               run4 = ph == CL_ALL || ph == 4;
    const unsigned tag = ((unsigned)*P.ao.serial << 7) | (unsigned)P.ao.slot;
    const KVT* kc = reinterpret_cast<const KVT*>(p.kv.k);
    const KVT* vc = reinterpret_cast<const KVT*>(p.kv.v);
    auto key_base = [&](int s) -> size_t {
        const int page = b * p.kv.pages_per_seq + (s >> 4);               // (contiguous page table: cp_layer_takes)
        return ((((size_t)p.layer * p.kv.n_pages + page) * p.kv.nkv + g) * 16 + (s & 15)) * HD;
    };
    // ---- requests at entry, in the order they are needed: the q|k|v strip's operands, the attention stage's, [no front: Wo], the gate|up block (DMA)
    cu32x4 gw[NKS], gx[NKS];
    if (run0) {
        const int nkt = P.ao.K / KT, kq = nkt >> 2;
        const cu32x4* wsrc = reinterpret_cast<const cu32x4*>(P.ao.Wqkv) + ((size_t)b_ * nkt + wave * kq) * 64 + lane;
        const cu32x4* xsrc = F32 ? reinterpret_cast<const cu32x4*>(reinterpret_cast<const float*>(P.ao.x16) + (size_t)(li < B ? li : 0) * P.ao.ldx16 + wave * kq * 16 + lq * 4)
                                 : reinterpret_cast<const cu32x4*>(P.ao.x16 + (size_t)(li < B ? li : 0) * P.ao.ldx16 + wave * kq * 32 + lq * 8);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { gw[ks] = wsrc[ks * 64]; gx[ks] = xsrc[ks * 4]; }
    }
    float xq[2] = {0.f, 0.f}, xk[2] = {0.f, 0.f}, xv[2] = {0.f, 0.f};
    float qw0 = 0.f, qw1 = 0.f, kw0 = 0.f, kw1 = 0.f, invf = 0.f, ctab = 0.f, stab = 0.f;
    const bool rtab = p.rope_cs && S0 < p.rope_cs_n;
    struct alignas(2 * sizeof(KVT)) VPair { KVT a, b; };
    cu32x4 kr[KW];
    VPair vr[MAXK];
    cu32x4 wf[2][NKS];
    auto load_wo = [&] {
        const int nkt = (p.nh * HD) / KT;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const cu32x4* wsrc = reinterpret_cast<const cu32x4*>(P.ao.Wo) + ((size_t)(c * 8 + wave * 2 + s) * nkt + g * NKS) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wf[s][ks] = wsrc[ks * 64];
        }
    };
    if (run1 && att_wg) {
        if constexpr (!QKV) {
            const float* xrow = p.qkv + (size_t)b * p.ld;
            xq[0] = xrow[(g * 2 + hh) * HD + lane]; xq[1] = xrow[(g * 2 + hh) * HD + lane + 64];
            xk[0] = xrow[(p.nh + g) * HD + lane]; xk[1] = xrow[(p.nh + g) * HD + lane + 64];
            xv[0] = xrow[(p.nh + p.nkv + g) * HD + lane]; xv[1] = xrow[(p.nh + p.nkv + g) * HD + lane + 64];
        }
        qw0 = p.qw[lane]; qw1 = p.qw[lane + 64]; kw0 = p.kw[lane]; kw1 = p.kw[lane + 64]; invf = p.inv_freq[lane];
        const float* rrow = rtab ? p.rope_cs + (size_t)S0 * 128 : p.inv_freq;
        ctab = rrow[lane]; stab = rrow[rtab ? 64 + lane : lane];
        const cu32x4* ksrc = reinterpret_cast<const cu32x4*>(kc + key_base(kk < S0 ? kk : 0) + qq * 32);
#pragma unroll
        for (int w = 0; w < KW; ++w) kr[w] = ksrc[w];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) vr[k] = *reinterpret_cast<const VPair*>(vc + key_base(k < S0 ? k : 0) + 2 * lane);
        if constexpr (!QKV) load_wo();
    }
    // the workgroup's block of the gate|up operator -> LDS, each wave the k quarter it will multiply (read back by the same wave only)
    constexpr int GU_WAVE = KQ * 4 * 2 * ACT * 16;      // bytes per wave
    static_assert(F32 || GU_WAVE % 1024 == 0, "cp_layer: a wave's gate|up block must be whole 1-KiB DMA requests");
    // `pace` (x 64 clocks between requests): 224 workgroups requesting 11 MB at one moment is a burst the reducers' small reads queue behind
    auto dma_gu = [&](int pace) {
        if constexpr (!F32) {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(P.mlp.Wgu) + ((size_t)b_ * 4 + wave) * GU_WAVE + lane * 16;
#pragma unroll
            for (int i = 0; i < GU_WAVE / 1024; ++i) {
                cl_dma16(src + i * 1024, gu_lds + wave * GU_WAVE + i * 1024);
                if (pace > 0) wt_first_pause(pace);
            }
        }
    };
    // when: 2 (default) -- behind the attention stage (after the partial o-projection is published / the hidden rows are: the workgroup then
    // waits >= 1.6 us for the hidden rows, longer than the block takes to arrive); requested at entry (0) or behind the o-projection's
    // operator (1) the 12.6 MB of a launch compete with the strips and with Wo: the attention stage ran 1.5-2 us late (profiles/r06_cp_layer.md)
    // The REDUCERS (32 of the 256 workgroups; everybody's phase B waits for their phase A) request theirs behind the o-projection's operator: their
    // 1.5 MB are no burst, and their way from the hidden rows' publication to their own phase A stays free.
    const int gu_when = (ph == CL_ALL && att_wg) ? ((P.gu_when == 2 && reducer) ? (QKV ? 1 : 0) : P.gu_when) : 0;
    if (run2 && gu_when == 0) dma_gu(0);
    const int done = p.done_flag ? *p.done_flag : 0;
    if (done) return;
    // ================================================================================================ stage 0: the q|k|v strip
    if (run0) {
        f32x4 qa = (f32x4){0.f, 0.f, 0.f, 0.f};
        float ssq = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            cu32x4 xv4 = gx[ks];
            if (li >= B) xv4 = (cu32x4){0u, 0u, 0u, 0u};
            if constexpr (F32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xe = __uint_as_float(xv4[e]);
                    ssq += xe * xe;
                    qa = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(gw[ks][e]), xe, qa, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(xv4[e] << 16), hi = __uint_as_float(xv4[e] & 0xffff0000u);
                    ssq += lo * lo; ssq += hi * hi;
                }
                bf16x8 wa, xb;
                *reinterpret_cast<cu32x4*>(&wa) = gw[ks];
                *reinterpret_cast<cu32x4*>(&xb) = xv4;
                qa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, qa, 0, 0, 0);
            }
        }
        ssq += __shfl_xor(ssq, 16);
        ssq += __shfl_xor(ssq, 32);
        f32x4* qpart = reinterpret_cast<f32x4*>(att_lds + L::WS + L::BT + L::OWN);
        float* qss = reinterpret_cast<float*>(att_lds + L::WS + L::BT + L::OWN + 4 * 64 * 16);
        qpart[wave * 64 + lane] = qa;
        if (lq == 0) qss[wave * 16 + li] = ssq;
        __syncthreads();
        if (wave == 0 && li < B) {
            const f32x4 s4 = ((qpart[lane] + qpart[64 + lane]) + qpart[128 + lane]) + qpart[192 + lane];
            const float ss = ((qss[li] + qss[16 + li]) + qss[32 + li]) + qss[48 + li];
            const float rs = rsqrtf(ss / (float)P.ao.K + P.ao.eps_in);
            const WtBuf qg = wt_buf(P.ao.qkv_gran, (size_t)8 * p.ld * 8);
            const int off = (int)(((size_t)li * p.ld + b_ * 16 + lq * 4) * 8);
            wt_store16(qg, off, (cu32x4){__float_as_uint(s4[0] * rs), tag, __float_as_uint(s4[1] * rs), tag});
            wt_store16(qg, off + 16, (cu32x4){__float_as_uint(s4[2] * rs), tag, __float_as_uint(s4[3] * rs), tag});
        }
    }
    // ================================================================================================ stage 1: attention + partial o-projection
    float x1a = 0.f, x1b = 0.f;                         // the reducer's hidden values after attention (threads < 128: row rq * 2 + tid / 64, columns c * 128 + 2 (tid % 64) + {0, 1})
    if (run1 && att_wg) {
        if constexpr (QKV) {
            load_wo();                                 // arrives while this workgroup waits for its rows and attends
            if (gu_when == 1) dma_gu(0);
            const WtBuf qg = wt_buf(P.ao.qkv_gran, (size_t)8 * p.ld * 8);
            int cols[3] = {(g * 2 + hh) * HD, (p.nh + g) * HD, (p.nh + p.nkv + g) * HD};
            uint2 gq[3][2], gn[3][2];
            auto load_rows = [&](uint2 (&d)[3][2]) {
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) d[v][h2] = wt_load8(qg, (int)(((size_t)b * p.ld + cols[v] + lane + 64 * h2) * 8));
            };
            wt_first_pause(P.ao.first_pause);
            load_rows(gq);
            wt_first_pause(P.ao.poll_step);
            load_rows(gn);
            for (int spins = 0;; ++spins) {
                bool fresh = true;
#pragma unroll
                for (int v = 0; v < 3; ++v) fresh = fresh && gq[v][0].y == tag && gq[v][1].y == tag;
                if (fresh) break;
                if (spins > GRANULE_SPIN_LIMIT) { cl_give_up(P); break; }
#pragma unroll
                for (int v = 0; v < 3; ++v) { gq[v][0] = gn[v][0]; gq[v][1] = gn[v][1]; }
                wt_first_pause(P.ao.poll_step);
                load_rows(gn);
            }
            xq[0] = __uint_as_float(gq[0][0].x); xq[1] = __uint_as_float(gq[0][1].x);
            xk[0] = __uint_as_float(gq[1][0].x); xk[1] = __uint_as_float(gq[1][1].x);
            xv[0] = __uint_as_float(gq[2][0].x); xv[1] = __uint_as_float(gq[2][1].x);
        }
        float* ws = reinterpret_cast<float*>(att_lds + wave * 1536);          // q | kn | vn
        bf16_t* Bt = reinterpret_cast<bf16_t*>(att_lds + L::WS);
        float* Btf = reinterpret_cast<float*>(att_lds + L::WS);
        float* own = reinterpret_cast<float*>(att_lds + L::WS + L::BT);
        if (have) {
            // ---- q / k RMSNorm + RoPE at position S0, K / V through the cache type (attn_cp's stage 1, as cp_attn_o_kernel)
            float c_ = ctab, sn = stab;
            if (!rtab) { const float ang = (float)S0 * invf; c_ = cosf(ang); sn = sinf(ang); }
            auto norm_rope = [&](float& x0, float& x1, float w0, float w1) {
                const float ss = wave_sum64_dpp(x0 * x0 + x1 * x1);
                const float rs = rsqrtf(ss / (float)HD + p.eps);
                x0 = w0 * (x0 * rs);
                x1 = w1 * (x1 * rs);
                const float o0 = x0 * c_ - x1 * sn, o1 = x1 * c_ + x0 * sn;
                x0 = o0; x1 = o1;
            };
            norm_rope(xq[0], xq[1], qw0, qw1);
            ws[lane] = xq[0]; ws[lane + 64] = xq[1];
            norm_rope(xk[0], xk[1], kw0, kw1);
            {
                const size_t o = key_base(S0);
                const KVT k0v = kv_cast<KVT>(xk[0]), k1v = kv_cast<KVT>(xk[1]), v0v = kv_cast<KVT>(xv[0]), v1v = kv_cast<KVT>(xv[1]);
                if (c == 0) {                              // one workgroup per (row pair, kv head) appends: head 0's wave the K row, head 1's the V row
                    if (hh == 0) { KVT* kd = reinterpret_cast<KVT*>(p.kv.k); kd[o + lane] = k0v; kd[o + lane + 64] = k1v; }
                    else { KVT* vd = reinterpret_cast<KVT*>(p.kv.v); vd[o + lane] = v0v; vd[o + lane + 64] = v1v; }
                }
                ws[HD + lane] = kv_load(&k0v); ws[HD + lane + 64] = kv_load(&k1v);
                ws[2 * HD + lane] = kv_load(&v0v); ws[2 * HD + lane + 64] = kv_load(&v1v);
            }
            __builtin_amdgcn_wave_barrier();
            const float* kn = ws + HD;
            const float* vn = ws + 2 * HD;
            float kx[32];
            if (kk == S0) {
#pragma unroll
                for (int e = 0; e < 32; ++e) kx[e] = kn[qq * 32 + e];
            } else {
#pragma unroll
                for (int w = 0; w < KW; ++w)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if constexpr (F32) kx[w * 4 + e] = __uint_as_float(kr[w][e]);
                        else {
                            kx[w * 8 + 2 * e] = __uint_as_float(kr[w][e] << 16);
                            kx[w * 8 + 2 * e + 1] = __uint_as_float(kr[w][e] & 0xffff0000u);
                        }
                    }
            }
            const float* q = ws + qq * 32;
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 32; ++e) a += q[e] * kx[e];
            a += __shfl_xor(a, 1);
            a += __shfl_xor(a, 2);
            const float s = kk < S1 ? a * rsqrtf((float)HD) : -INFINITY;
            const float m = wave_max64_dpp(s);
            const float e = kk < S1 ? att_exp<KVT>(s - m) : 0.f;
            const float l = wave_sum64_dpp(qq == 0 ? e : 0.f);
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int k = 0; k < MAXK; ++k) {
                if (k < S0) {
                    const float ek = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(e), k * 4));
                    acc0 += ek * kv_load(&vr[k].a);
                    acc1 += ek * kv_load(&vr[k].b);
                }
            }
            {
                const float ek = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(e), S0 * 4));
                acc0 += ek * vn[2 * lane];
                acc1 += ek * vn[2 * lane + 1];
            }
            const float inv = 1.f / l;
            if constexpr (F32) {
                float2 o2; o2.x = acc0 * inv; o2.y = acc1 * inv;
                *reinterpret_cast<float2*>(Btf + rr * BSTR + hh * HD + 2 * lane) = o2;
            } else {
                const unsigned pk = (unsigned)f32_to_bf16(acc0 * inv) | ((unsigned)f32_to_bf16(acc1 * inv) << 16);
                *reinterpret_cast<unsigned*>(Bt + rr * BSTR + hh * HD + 2 * lane) = pk;
            }
        }
        __syncthreads();
        // ---- partial o-projection: D[feature 4 q + j][sequence li] of two strips over the 256 k of this kv head (columns 0 / 1 of the MFMA tile)
        const int row = rq * 2 + li;                        // (meaningful for li < 2)
        const bool col_ok = li < 2 && row < B;
        f32x4 acc[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            cu32x4 bv = F32 ? *reinterpret_cast<const cu32x4*>(Btf + (li & 1) * BSTR + ks * 16 + lq * 4)
                            : *reinterpret_cast<const cu32x4*>(Bt + (li & 1) * BSTR + ks * 32 + lq * 8);
            if (!col_ok) bv = (cu32x4){0u, 0u, 0u, 0u};
            if constexpr (F32) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[s][ks][e]), __uint_as_float(bv[e]), acc[s], 0, 0, 0);
            } else {
                bf16x8 xb;
                *reinterpret_cast<cu32x4*>(&xb) = bv;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 wa;
                    *reinterpret_cast<cu32x4*>(&wa) = wf[s][ks];
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[s], 0, 0, 0);
                }
            }
        }
        const WtBuf slab = wt_buf(P.ao.part, (size_t)NKV * 8 * H * 8);
        if (reducer) {
            if (li < 2) {
#pragma unroll
                for (int s = 0; s < 2; ++s) *reinterpret_cast<f32x4*>(own + li * 128 + (wave * 2 + s) * 16 + lq * 4) = acc[s];
            }
        } else if (col_ok) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int off = (int)((((size_t)g * 8 + row) * H + c * 128 + (wave * 2 + s) * 16 + lq * 4) * 8);
                wt_store16(slab, off, (cu32x4){__float_as_uint(acc[s][0]), tag, __float_as_uint(acc[s][1]), tag});
                wt_store16(slab, off + 16, (cu32x4){__float_as_uint(acc[s][2]), tag, __float_as_uint(acc[s][3]), tag});
            }
        }
        QTTS_TS(1);
        if (gu_when == 1 && !QKV) dma_gu(0);
        if (gu_when == 2) dma_gu(P.gu_pace);
        // ============================================================================================ stage 2: the o-projection's reducers -> hidden rows
        if (reducer) {
            __syncthreads();                                 // (its own partial sum is in LDS)
            if (tid < 128) {
                const int rw = rq * 2 + (tid >> 6), c2 = (tid & 63) * 2, col = c * 128 + c2;
                if (rw < B) {
                    const float2 res = *reinterpret_cast<const float2*>(P.ao.res + (size_t)rw * H + col);
                    cu32x4 pa[NKV - 1], pn[NKV - 1];
                    auto load_slabs = [&](cu32x4 (&d)[NKV - 1]) {
#pragma unroll
                        for (int g2 = 0; g2 < NKV - 1; ++g2) d[g2] = wt_load16(slab, (int)((((size_t)g2 * 8 + rw) * H + col) * 8));
                    };
                    wt_first_pause(P.ao.first_pause);
                    load_slabs(pa);
                    wt_first_pause(P.ao.poll_step);
                    load_slabs(pn);
                    for (int spins = 0;; ++spins) {
                        bool fresh = true;
#pragma unroll
                        for (int g2 = 0; g2 < NKV - 1; ++g2) fresh = fresh && pa[g2][1] == tag && pa[g2][3] == tag;
                        if (fresh) break;
                        if (spins > GRANULE_SPIN_LIMIT) { cl_give_up(P); break; }
#pragma unroll
                        for (int g2 = 0; g2 < NKV - 1; ++g2) pa[g2] = pn[g2];
                        wt_first_pause(P.ao.poll_step);
                        load_slabs(pn);
                    }
                    float s0 = __uint_as_float(pa[0][0]), s1 = __uint_as_float(pa[0][2]);
#pragma unroll
                    for (int g2 = 1; g2 < NKV - 1; ++g2) { s0 += __uint_as_float(pa[g2][0]); s1 += __uint_as_float(pa[g2][2]); }
                    s0 += own[(tid >> 6) * 128 + c2]; s1 += own[(tid >> 6) * 128 + c2 + 1];
                    s0 += res.x; s1 += res.y;
                    x1a = s0; x1b = s1;
                    // the hidden rows after attention: to every workgroup's phase A as granules -- {2 x bf16, tag} (fp32 engines: {fp32, tag} x 2)
                    const int hid_stride = 8 * (F32 ? H : H / 2) * 8;
                    const WtBuf hg = wt_buf(reinterpret_cast<unsigned char*>(P.hid_gran) + (size_t)P.hid_slot * hid_stride, (size_t)hid_stride);
                    if constexpr (F32) {
                        wt_store16(hg, (int)(((size_t)rw * H + col) * 8), (cu32x4){__float_as_uint(s0), tag, __float_as_uint(s1), tag});
                    } else {
#ifdef QTTS_HOST_EMU
                        *reinterpret_cast<uint2*>(hg.base + ((size_t)rw * (H / 2) + (col >> 1)) * 8) = uint2{pack_bf16(s0, s1), tag};
#else
                        typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));
                        __builtin_amdgcn_raw_buffer_store_b64((cu32x2){pack_bf16(s0, s1), tag}, hg.r, (int)(((size_t)rw * (H / 2) + (col >> 1)) * 8), 0, 16);
#endif
                    }
                    QTTS_TS(1);                              // (reducers: stamp 1 = the hidden rows published)
                    if (ph != CL_ALL) {                      // (emulated stages: the residual of stage 5 travels through the output rows)
                        float2 o2; o2.x = s0; o2.y = s1;
                        *reinterpret_cast<float2*>(P.mlp.out + (size_t)rw * H + col) = o2;
                    }
                }
            }
        }
    }
    if (!run2 && !run3 && !run4) return;
    // ================================================================================================ stage 3: phase A (gate|up over the hidden rows)
    constexpr int KTM = F32 ? 16 : 32;
    const int slice = I >> 3;
    const int spairs = F32 ? slice : slice >> 1;
    f32x4* qa_l = reinterpret_cast<f32x4*>(mlp_lds);
    float* qss_l = reinterpret_cast<float*>(mlp_lds + 4 * 64 * 2 * 16);
    f32x4* qb_l = reinterpret_cast<f32x4*>(mlp_lds + L::QA);
    const WtBuf ag = wt_buf(P.mlp.act_gran, (size_t)8 * 8 * spairs * 8);
    const int nktI = I / KTM;
    cu32x4 wd[2][KTW];
    auto load_wd = [&] {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const cu32x4* dsrc = reinterpret_cast<const cu32x4*>(P.mlp.Wd) + ((size_t)(j * 2 + t2) * nktI + xcd * (slice / KTM) + wave * KTW) * 64 + lane;
#pragma unroll
            for (int t = 0; t < KTW; ++t) wd[t2][t] = dsrc[t * 64];
        }
    };
    if (run2) {
        if (ph == CL_ALL) __syncthreads();                 // the attention stage's LDS (B tile, the reducer's own partial sum) is dead: phase A's quarters overwrite it
        const int row = li < B ? li : 0;
        // this wave's k quarter of the hidden rows: per k-tile 8 bf16 = 4 granules = 32 B per lane (fp32: 4 values = 4 granules = 32 B)
        cu32x4 wgr[F32 ? KQ : 1], wur[F32 ? KQ : 1];     // fp32 engines: the gate|up tiles in registers, requested here
        if constexpr (F32) {
            const cu32x4* wsrc = reinterpret_cast<const cu32x4*>(P.mlp.Wgu) + (((size_t)b_ * (H / KTM) + wave * KQ) * 4 + lq) * (2 * ACT) + (li < ACT ? li : 0);
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks) { wgr[ks] = wsrc[(size_t)ks * 4 * 2 * ACT]; wur[ks] = wsrc[(size_t)ks * 4 * 2 * ACT + ACT]; }
        }
        if (run3) load_wd();                             // 24 KB: streams while the hidden rows travel
        // The hidden rows reach 1024 waves from 32 reducers.  First version: every wave polled its whole k quarter (8 KB per wave and round =
        // 8 MB per round through the fabric) -- the hand-off took 3.7 us in the kernel's timeline.  Now a wave polls one SENTINEL granule per
        // producer wave it depends on (the first granule of each (row, 128-feature chunk): 16 lanes x 8 B), and reads its quarter once the
        // sentinels carry the tag -- every granule's own tag is still checked, and re-read (sc1) until fresh: a producer wave's 64 granules
        // leave in ONE store instruction, so they become readable together, but nothing orders them.
        // hid_mode 2: that one read may be served by this XCD's L2 (the region of the buffer is this launch slot's own: nobody has read it
        // since the previous frame): 32 workgroups of an XCD pull each line through the fabric once instead of 32 times.
        const int hid_stride = 8 * (F32 ? H : H / 2) * 8;                 // bytes of one launch slot's region
        const WtBuf hgs = wt_buf(reinterpret_cast<unsigned char*>(P.hid_gran) + (size_t)P.hid_slot * hid_stride, (size_t)hid_stride);
        int hoff[KQ];
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks)
            hoff[ks] = F32 ? (int)(((size_t)row * H + (wave * KQ + ks) * 16 + lq * 4) * 8) : (int)(((size_t)row * (H / 2) + (wave * KQ + ks) * 16 + lq * 4) * 8);
        constexpr int NV = KQ * KTM, NCH = (NV + 127) / 128;              // values / 128-feature chunks of this wave's k quarter
        const int srow = lane / NCH, scc = lane - srow * NCH;
        const bool s_live = lane < 8 * NCH && srow < B;
        const int c_s = (wave * NV) / 128 + scc;
        const int soff = F32 ? (int)(((size_t)(s_live ? srow : 0) * H + c_s * 128) * 8) : (int)(((size_t)(s_live ? srow : 0) * (H / 2) + c_s * 64) * 8);
        cu32x4 cur[KQ][2];
        auto load_hid = [&](cu32x4 (&d)[KQ][2]) {
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks) { d[ks][0] = wt_load16(hgs, hoff[ks]); d[ks][1] = wt_load16(hgs, hoff[ks] + 16); }
        };
        auto load_hid_cached = [&](cu32x4 (&d)[KQ][2]) {
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks) { d[ks][0] = wt_load16_cached(hgs, hoff[ks]); d[ks][1] = wt_load16_cached(hgs, hoff[ks] + 16); }
        };
        auto all_fresh = [&](const cu32x4 (&d)[KQ][2]) {
            bool fresh = true;
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) fresh = fresh && d[ks][h2][1] == tag && d[ks][h2][3] == tag;
            return fresh;
        };
        // (bf16) the gate|up tiles: LDS -> registers while the first read of the hidden rows is in flight -- the DMA requests are older than that
        // read, so "all but the newest 2 KQ requests" covers them; phase A then runs from registers as cp_mlp_kernel's does (second timeline:
        // with the LDS reads inside the MFMA loop, hidden rows in -> act published took 1.5 us against cp_mlp's 0.5)
        cu32x4 wg_l[F32 ? 1 : KQ], wu_l[F32 ? 1 : KQ];
        auto preload_gu = [&] {
            if constexpr (!F32) {
#pragma unroll
                for (int ks = 0; ks < KQ; ++ks) {
                    const unsigned char* wl = gu_lds + ((size_t)((wave * KQ + ks) * 4 + lq) * (2 * ACT) + (li < ACT ? li : 0)) * 16;
                    wg_l[ks] = *reinterpret_cast<const cu32x4*>(wl);
                    wu_l[ks] = *reinterpret_cast<const cu32x4*>(wl + ACT * 16);
                }
            }
        };
        wt_first_pause(P.pause_h);
        if (P.hid_mode != 0) {
            uint2 sv = wt_load8(hgs, soff);
            wt_first_pause(P.mlp.poll_step);
            uint2 sn = wt_load8(hgs, soff);
            for (int spins = 0;; ++spins) {
                if (__ballot(s_live && sv.y != tag) == 0ull) break;
                if (spins > GRANULE_SPIN_LIMIT) { cl_give_up(P); break; }
                sv = sn;
                wt_first_pause(P.mlp.poll_step);
                sn = wt_load8(hgs, soff);
            }
            if (P.hid_mode == 2) load_hid_cached(cur); else load_hid(cur);
        } else load_hid(cur);
        if constexpr (!F32) { cl_dma_wait<2 * KQ>(); preload_gu(); }
        for (int spins = 0;; ++spins) {                   // (hid_mode 0: the polling loop; otherwise it runs through once unless a granule lags its sentinel)
            if (all_fresh(cur)) break;
            if (spins > GRANULE_SPIN_LIMIT) { cl_give_up(P); break; }
            wt_first_pause(P.mlp.poll_step);
            load_hid(cur);
        }
        QTTS_TS(2);
        f32x4 ag4 = (f32x4){0.f, 0.f, 0.f, 0.f}, au4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        float ssq = 0.f;
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) {
            cu32x4 xv4 = (cu32x4){cur[ks][0][0], cur[ks][0][2], cur[ks][1][0], cur[ks][1][2]};
            if (li >= B) xv4 = (cu32x4){0u, 0u, 0u, 0u};
            if constexpr (F32) {
                cu32x4 g4 = wgr[ks], u4 = wur[ks];
                if (li >= ACT) { g4 = (cu32x4){0u, 0u, 0u, 0u}; u4 = g4; }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xe = __uint_as_float(xv4[e]);
                    ssq += xe * xe;
                    ag4 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(g4[e]), xe, ag4, 0, 0, 0);
                    au4 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(u4[e]), xe, au4, 0, 0, 0);
                }
            } else {
                cu32x4 g4 = wg_l[ks], u4 = wu_l[ks];
                if (li >= ACT) { g4 = (cu32x4){0u, 0u, 0u, 0u}; u4 = g4; }
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
        ssq += __shfl_xor(ssq, 16);
        ssq += __shfl_xor(ssq, 32);
        qa_l[(wave * 64 + lane) * 2] = ag4;
        qa_l[(wave * 64 + lane) * 2 + 1] = au4;
        if (lq == 0) qss_l[wave * 16 + li] = ssq;
        __syncthreads();
        if (wave == 0 && li < B && lq * 4 < ACT) {
            f32x4 sg = ((qa_l[lane * 2] + qa_l[(64 + lane) * 2]) + qa_l[(128 + lane) * 2]) + qa_l[(192 + lane) * 2];
            f32x4 su = ((qa_l[lane * 2 + 1] + qa_l[(64 + lane) * 2 + 1]) + qa_l[(128 + lane) * 2 + 1]) + qa_l[(192 + lane) * 2 + 1];
            const float ss = ((qss_l[li] + qss_l[16 + li]) + qss_l[32 + li]) + qss_l[48 + li];
            const float rs = rsqrtf(ss / (float)H + P.mlp.eps);
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
        QTTS_TS(3);
    } else if (run3) load_wd();
    if (!run3 && !run4) return;
    // ================================================================================================ stage 4: phase B (partial down-projection over the XCD's slice)
    const WtBuf mslab = wt_buf(P.mlp.part, (size_t)8 * 8 * H * 8);
    if (run3) {
        const int r_t = tid >> 5, f_t = tid & 31;
        const int row = li < B ? li : 0;
        int offs[KTW];
#pragma unroll
        for (int t = 0; t < KTW; ++t) offs[t] = (int)((((size_t)xcd * 8 + row) * spairs + (wave * KTW + t) * 16 + lq * 4) * 8);
        cu32x4 cur[KTW][2], nxt[KTW][2];
        auto load_slice = [&](cu32x4 (&d)[KTW][2]) {
#pragma unroll
            for (int t = 0; t < KTW; ++t) { d[t][0] = wt_load16(ag, offs[t]); d[t][1] = wt_load16(ag, offs[t] + 16); }
        };
        wt_first_pause(P.mlp.first_pause);
        load_slice(cur);
        wt_first_pause(P.mlp.poll_step);
        load_slice(nxt);
        for (int spins = 0;; ++spins) {
            bool fresh = true;
#pragma unroll
            for (int t = 0; t < KTW; ++t)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) fresh = fresh && cur[t][h2][1] == tag && cur[t][h2][3] == tag;
            if (fresh) break;
            if (spins > GRANULE_SPIN_LIMIT) { cl_give_up(P); break; }
#pragma unroll
            for (int t = 0; t < KTW; ++t) { cur[t][0] = nxt[t][0]; cur[t][1] = nxt[t][1]; }
            wt_first_pause(P.mlp.poll_step);
            load_slice(nxt);
        }
        f32x4 acc[2];
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
        for (int t = 0; t < KTW; ++t) {
            cu32x4 xv4 = (cu32x4){cur[t][0][0], cur[t][0][2], cur[t][1][0], cur[t][1][2]};
            if (li >= B) xv4 = (cu32x4){0u, 0u, 0u, 0u};
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
        qb_l[(wave * 2 + 0) * 64 + lane] = acc[0];
        qb_l[(wave * 2 + 1) * 64 + lane] = acc[1];
        __syncthreads();
        const float* qf = reinterpret_cast<const float*>(qb_l);
        const int e = (((f_t >> 4) * 64 + ((f_t & 15) >> 2) * 16 + r_t) << 2) + (f_t & 3);
        const float mine = ((qf[e] + qf[2 * 64 * 4 + e]) + qf[4 * 64 * 4 + e]) + qf[6 * 64 * 4 + e];
        if (r_t < B) {
            const int off = (int)((((size_t)xcd * 8 + r_t) * H + j * 32 + f_t) * 8);
#ifdef QTTS_HOST_EMU
            *reinterpret_cast<uint2*>(mslab.base + off) = uint2{__float_as_uint(mine), tag};
#else
            typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64((cu32x2){__float_as_uint(mine), tag}, mslab.r, off, 0, 16);
#endif
        }
        QTTS_TS(4);
    }
    if (!run4 || !reducer) { QTTS_TS_CPLAYER(0) return; }
    // ================================================================================================ stage 5: the MLP's reducers: 8 XCD partials in XCD order + x1
    if (tid < 128) {
        const int rw = rq * 2 + (tid >> 6), c2 = (tid & 63) * 2, col = c * 128 + c2;
        if (rw < B) {
            if (ph != CL_ALL) { const float2 r2 = *reinterpret_cast<const float2*>(P.mlp.out + (size_t)rw * H + col); x1a = r2.x; x1b = r2.y; }
            constexpr int NP = 8;
            cu32x4 pa[NP], pn[NP];
            auto load_slabs = [&](cu32x4 (&d)[NP]) {
#pragma unroll
                for (int x2 = 0; x2 < NP; ++x2) d[x2] = wt_load16(mslab, (int)((((size_t)x2 * 8 + rw) * H + col) * 8));
            };
            wt_first_pause(P.mlp.pause_c);
            load_slabs(pa);
            wt_first_pause(P.mlp.poll_step);
            load_slabs(pn);
            for (int spins = 0;; ++spins) {
                bool fresh = true;
#pragma unroll
                for (int x2 = 0; x2 < NP; ++x2) fresh = fresh && pa[x2][1] == tag && pa[x2][3] == tag;
                if (fresh) break;
                if (spins > GRANULE_SPIN_LIMIT) { cl_give_up(P); break; }
#pragma unroll
                for (int x2 = 0; x2 < NP; ++x2) pa[x2] = pn[x2];
                wt_first_pause(P.mlp.poll_step);
                load_slabs(pn);
            }
            float s0 = __uint_as_float(pa[0][0]), s1 = __uint_as_float(pa[0][2]);
#pragma unroll
            for (int x2 = 1; x2 < NP; ++x2) { s0 += __uint_as_float(pa[x2][0]); s1 += __uint_as_float(pa[x2][2]); }
            s0 += x1a; s1 += x1b;
            float2 o2; o2.x = s0; o2.y = s1;
            *reinterpret_cast<float2*>(P.mlp.out + (size_t)rw * H + col) = o2;
            if (P.mlp.out16) *reinterpret_cast<unsigned*>(P.mlp.out16 + (size_t)rw * H + col) = pack_bf16(s0, s1);
        }
    }
    QTTS_TS_DRAINED(5);
    QTTS_TS_CPLAYER(1)
}

// ------------------------------------------------------------------------------------------------------------------------ host side
bool cp_layer_takes(const AttnDecodeParams& a, int H, int I) {
    return cp_attn_o_takes(a, H) && a.kv.contig && cp_mlp_takes(a.B, H, I) && cp_attn_o_grid(H) == cp_mlp_grid(H);
}
int cp_layer_grid(int H) { return cp_mlp_grid(H); }

static thread_local hipEvent_t tl_cl_ev_start = nullptr, tl_cl_ev_stop = nullptr;
void cp_layer_set_launch_events(hipEvent_t start, hipEvent_t stop) { tl_cl_ev_start = start; tl_cl_ev_stop = stop; }

// (bf16: k-tiles of 32, KQ = H / 128, KTW = I / 1024; fp32: k-tiles of 16, KQ = H / 64, KTW = I / 512)
#define QTTS_CPLAYER_CASES(X) X(false, 12, 8, 3) X(false, 16, 2, 1) X(true, 12, 16, 6) X(true, 16, 4, 2)

static bool cp_layer_shape(int H, int I, bool f32, int& act, int& kq, int& ktw) {
    if (!cp_mlp_takes(1, H, I)) return false;
    act = I / (H / 4); kq = H / (f32 ? 64 : 128); ktw = I / 8 / (f32 ? 16 : 32) / 4;
    return true;
}

template <bool QKV, bool F32, int ACT, int KQ, int KTW>
static void launch_cp_layer_t(const CpLayerParams& P, hipStream_t st) {
    const dim3 grid(cp_layer_grid(P.ao.H));
    auto kern = cp_layer_kernel<QKV, F32, ACT, KQ, KTW>;
    constexpr int lds = ClLds<QKV, F32, ACT, KQ>::TOTAL;
#ifdef QTTS_HOST_EMU
    for (int ph = QKV ? 0 : 1; ph < 5; ++ph) {
        if (P.phase != CL_ALL && P.phase != ph) continue;       // (a single stage given: that stage alone -- the stale-granule tests)
        CpLayerParams Q = P;
        Q.phase = ph;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, QTTS_CPLAYER_ARGS(Q));
    }
#else
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (tl_cl_ev_start) hipExtLaunchKernelGGL(kern, grid, dim3(256), lds, st, tl_cl_ev_start, tl_cl_ev_stop, 0, QTTS_CPLAYER_ARGS(P));
    else hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, QTTS_CPLAYER_ARGS(P));
#endif
}

bool cp_layer_instantiated(int H, int I, bool bf16) {
    int act = 0, kq = 0, ktw = 0;
    if (!cp_layer_shape(H, I, !bf16, act, kq, ktw)) return false;
    const bool f32 = !bf16;
#define QTTS_CPLAYER_X(F, A, Q, T) if (f32 == F && act == A && kq == Q && ktw == T) return true;
    QTTS_CPLAYER_CASES(QTTS_CPLAYER_X)
#undef QTTS_CPLAYER_X
    return false;
}

int cp_layer_lds_bytes(int H, int I, bool bf16) {
    int act = 0, kq = 0, ktw = 0;
    if (!cp_layer_shape(H, I, !bf16, act, kq, ktw)) return 0;
    const bool f32 = !bf16;
#define QTTS_CPLAYER_X(F, A, Q, T) if (f32 == F && act == A && kq == Q && ktw == T) return ClLds<true, F, A, Q>::TOTAL;
    QTTS_CPLAYER_CASES(QTTS_CPLAYER_X)
#undef QTTS_CPLAYER_X
    return 0;
}

int cp_layer_blocks_per_cu(int H, int I, bool bf16) {
#ifdef QTTS_HOST_EMU
    if (const char* e = QTTS_ENV("QTTS_HOSTEMU_CPAO_BLOCKS_PER_CU")) return atoi(e);
    return bf16 ? 2 : 1;
#else
    int act = 0, kq = 0, ktw = 0;
    if (!cp_layer_shape(H, I, !bf16, act, kq, ktw)) return 0;
    const bool f32 = !bf16;
    int best = 1 << 30;
#define QTTS_CPLAYER_X(F, A, Q, T)                                                                                                          \
    if (f32 == F && act == A && kq == Q && ktw == T) {                                                                                      \
        int n = 0;                                                                                                                          \
        ensure_dynamic_lds(reinterpret_cast<const void*>(cp_layer_kernel<true, F, A, Q, T>), ClLds<true, F, A, Q>::TOTAL);                  \
        QTTS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cp_layer_kernel<true, F, A, Q, T>, 256, ClLds<true, F, A, Q>::TOTAL)); \
        best = std::min(best, n);                                                                                                           \
        ensure_dynamic_lds(reinterpret_cast<const void*>(cp_layer_kernel<false, F, A, Q, T>), ClLds<false, F, A, Q>::TOTAL);                \
        QTTS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cp_layer_kernel<false, F, A, Q, T>, 256, ClLds<false, F, A, Q>::TOTAL)); \
        best = std::min(best, n);                                                                                                           \
        return best;                                                                                                                        \
    }
    QTTS_CPLAYER_CASES(QTTS_CPLAYER_X)
#undef QTTS_CPLAYER_X
    return 0;
#endif
}

void launch_cp_layer(const CpLayerParams& P, hipStream_t st) {
    QTTS_REQUIRE(cp_layer_takes(P.ao.a, P.ao.H, P.mlp.I) && P.mlp.H == P.ao.H && P.mlp.B == P.ao.a.B, QTTS_ERR_ARG,
                 "cp_layer: shape (16 / 8 heads of 128, one new token, <= 16 keys, batch <= 8, contiguous pages, H % 128, I / (H / 4) in {4, 8, 12, 16})");
    const bool f32 = !P.ao.a.kv.bf16;
    QTTS_REQUIRE((P.mlp.f32 != 0) == f32 && (!f32 || !P.mlp.out16), QTTS_ERR_ARG, "cp_layer: operators, rows and cache of one type; no bf16 copy in fp32 mode");
    QTTS_REQUIRE(P.ao.Wo && P.ao.res && P.ao.part && P.ao.serial && P.ao.a.qw && P.ao.a.kw && P.ao.a.inv_freq && P.mlp.Wgu && P.mlp.Wd && P.mlp.out &&
                     P.mlp.act_gran && P.mlp.part && P.hid_gran, QTTS_ERR_ARG, "cp_layer: null operand");
    QTTS_REQUIRE(P.ao.slot >= 0 && P.ao.slot < 128, QTTS_ERR_ARG, "cp_layer: slot must be 0..127");
    const bool front = P.ao.Wqkv != nullptr;
    if (front)
        QTTS_REQUIRE(cp_layer_grid(P.ao.H) * 16 == P.ao.a.ld && P.ao.K == 1024 && P.ao.x16 && P.ao.qkv_gran && P.ao.ldx16 % 8 == 0, QTTS_ERR_ARG,
                     "cp_layer: the q|k|v front needs K = 1024, (nh + 2 nkv) * 128 == 16 * workgroups, x rows in the engine's type and the granule buffer");
    else QTTS_REQUIRE(P.ao.a.qkv, QTTS_ERR_ARG, "cp_layer: null q|k|v rows");
    int act = 0, kq = 0, ktw = 0;
    cp_layer_shape(P.ao.H, P.mlp.I, f32, act, kq, ktw);
#define QTTS_CPLAYER_X(F, A, Q, T)                                                                     \
    if (f32 == F && act == A && kq == Q && ktw == T) {                                                 \
        if (front) launch_cp_layer_t<true, F, A, Q, T>(P, st); else launch_cp_layer_t<false, F, A, Q, T>(P, st); \
        QTTS_CHECK_HIP(hipGetLastError());                                                             \
        return;                                                                                        \
    }
    QTTS_CPLAYER_CASES(QTTS_CPLAYER_X)
#undef QTTS_CPLAYER_X
    throw Error(QTTS_ERR_ARG, "cp_layer: no instantiation for this (H, I, dtype)");
}

}  // namespace qtts
