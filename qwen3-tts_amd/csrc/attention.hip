// attention.hip -- attention kernels for gfx950 (wave64).
//
//  attn_rows         one wave per (batch, head, query row) over a fused fp32 qkv buffer; causal, optional
//                    sliding window and left-pad mask.  Used by the talker prefill and by the codec's
//                    8-layer window-72 transformer.  16 lanes cooperate on one key (hd/16 dims per lane,
//                    16-B loads), 4 keys in flight per wave, two passes (max, then exp/sum/PV) so the
//                    softmax is the plain fp32 softmax of the reference (M:652, tokenizer v2:139).
//  qknorm_rope_store prefill: per-head RMSNorm (M:752-757) + rotate-half RoPE (M:660-724 == plain RoPE,
//                    SURVEY.md 3.2) in place on q,k and append of K,V to the paged cache.
//  attn_decode       decode: the same norm/RoPE/append fused with single-query GQA attention over the
//                    paged KV cache; one workgroup per (sequence, kv head) serves the whole q-head group so
//                    each K/V tile is read once; scores staged in LDS; fp32 softmax.
#include "common.h"
#include "kernels.h"
#include "tstamp.h"
#include "granule.h"
#include "attn_helpers.h"
#include <hip/hip_ext.h>

QTTS_TS_UNIT(attn)

namespace qtts {

__device__ inline float group16_sum(float v) { return row16_sum(v); }
__device__ inline float wave_sum64(float v) { return wave_sum64_dpp(v); }
__device__ inline float wave_max64(float v) { return wave_max64_dpp(v); }

// =================================================================================== attn_rows
template <int HD>
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnRowsParams p) {
    constexpr int DPL = HD / 16;  // dims per lane
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, li = lane & 15;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;   // (b, h, tq)
    const int64_t total = (int64_t)p.B * p.nh * p.T;
    if (item >= total) return;
    const int tq = (int)(item % p.T);
    const int h = (int)((item / p.T) % p.nh);
    const int b = (int)(item / ((int64_t)p.T * p.nh));
    const int npad = p.n_pad ? p.n_pad[b] : 0;
    const size_t ooff = ((size_t)b * p.T + tq) * p.ldo + h * HD + li * DPL;
    float* orow = p.out + ooff;
    bf16_t* orow16 = reinterpret_cast<bf16_t*>(p.out16) + ooff;
    if (tq < npad) {  // left-pad query row: never read downstream
        if (g == 0)
#pragma unroll
            for (int d = 0; d < DPL; ++d) { if (p.out16) orow16[d] = 0; else orow[d] = 0.f; }
        return;
    }
    const int kvh = h / (p.nh / p.nkv);
    const float scale = rsqrtf((float)HD);
    const float* base = p.qkv + (size_t)b * p.T * p.ld;
    float q[DPL];
    {
        const float* qp = base + (size_t)tq * p.ld + p.q_off + h * HD + li * DPL;
#pragma unroll
        for (int d = 0; d < DPL; ++d) q[d] = qp[d];
    }
    int lo = npad;
    if (p.window > 0 && tq - p.window + 1 > lo) lo = tq - p.window + 1;
    const int hi = tq;
    const float* kb = base + p.k_off + kvh * HD + li * DPL;
    const float* vb = base + p.v_off + kvh * HD + li * DPL;
    // Round 4: UN keys per 16-lane group and loop trip (16 per wave), their K (and V) rows requested together and unconditionally (keys past
    // `hi` re-read key `hi` and are skipped at use): the loop was one dependent L2 round trip per 4 keys -- 48 us per layer for the
    // batch-32 prefill's 50-key rows (profiles/r03_config4_kernel_trace.md).  Keys are still folded in ascending order per group, so the
    // sums are bit-identical to the one-key-per-trip loop.
    constexpr int UN = 4;
    // pass 1: row max
    float m = -INFINITY;
    for (int s0 = lo + g; s0 <= hi; s0 += 4 * UN) {
        float kr[UN][DPL];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int sc = s0 + 4 * u <= hi ? s0 + 4 * u : hi;
            const float* kp = kb + (size_t)sc * p.ld;
#pragma unroll
            for (int e = 0; e < DPL; ++e) kr[u][e] = kp[e];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (s0 + 4 * u > hi) continue;
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) d += q[e] * kr[u][e];
            d = group16_sum(d) * scale;
            m = fmaxf(m, d);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    // pass 2: exp, sum, PV
    float l = 0.f, acc[DPL];
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] = 0.f;
    for (int s0 = lo + g; s0 <= hi; s0 += 4 * UN) {
        float kr[UN][DPL], vr[UN][DPL];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int sc = s0 + 4 * u <= hi ? s0 + 4 * u : hi;
            const float* kp = kb + (size_t)sc * p.ld;
            const float* vp = vb + (size_t)sc * p.ld;
#pragma unroll
            for (int e = 0; e < DPL; ++e) { kr[u][e] = kp[e]; vr[u][e] = vp[e]; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (s0 + 4 * u > hi) continue;
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < DPL; ++e) d += q[e] * kr[u][e];
            d = group16_sum(d) * scale;
            const float pr = expf(d - m);
            l += pr;
#pragma unroll
            for (int e = 0; e < DPL; ++e) acc[e] += pr * vr[u][e];
        }
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
#pragma unroll
    for (int e = 0; e < DPL; ++e) {
        acc[e] += __shfl_xor(acc[e], 16);
        acc[e] += __shfl_xor(acc[e], 32);
    }
    if (g == 0) {
        const float inv = 1.f / l;
#pragma unroll
        for (int e = 0; e < DPL; ++e) { if (p.out16) orow16[e] = f32_to_bf16(acc[e] * inv); else orow[e] = acc[e] * inv; }
    }
}

// Round 6: R = 4 consecutive query rows of a (sequence, head) per wave, for the full-causal case (no sliding window: the talker's prefill).  A wave of
// attn_rows_kernel is a chain of dependent L2 round trips (16 keys per trip, two passes), and a batch-32 prefill is 32 768 of them per layer -- seven
// residency rounds: 91 us per layer, 2.6 ms of config 4's 12 ms prefill (profiles/r06_kernel_trace_frame_b32.md).  Here a trip's K (and V) rows serve
// four query rows: a quarter of the waves and of the K / V reads.  Per row NOTHING changes: without a window every row of a sequence starts at the same
// key (lo = n_pad), so key s belongs to the same 16-lane group, is folded in the same (ascending) order and the groups are combined by the same
// shuffles -- bit-identical to attn_rows_kernel row by row (emulator test), which is what the fp32 parity mode's goldens need.
template <int HD>
__global__ __launch_bounds__(256) void attn_rows4_kernel(AttnRowsParams p) {
    constexpr int DPL = HD / 16, R = 4, UN = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, li = lane & 15;
    const int nblk = (p.T + R - 1) / R;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;   // (b, h, row block)
    if (item >= (int64_t)p.B * p.nh * nblk) return;
    const int t0 = (int)(item % nblk) * R;
    const int h = (int)((item / nblk) % p.nh);
    const int b = (int)(item / ((int64_t)nblk * p.nh));
    const int npad = p.n_pad ? p.n_pad[b] : 0;
    const int kvh = h / (p.nh / p.nkv);
    const float scale = rsqrtf((float)HD);
    const float* base = p.qkv + (size_t)b * p.T * p.ld;
    float q[R][DPL];
    bool live[R];                                           // a real, non-pad query row
#pragma unroll
    for (int r = 0; r < R; ++r) {
        live[r] = t0 + r < p.T && t0 + r >= npad;
        const float* qp = base + (size_t)(t0 + r < p.T ? t0 + r : p.T - 1) * p.ld + p.q_off + h * HD + li * DPL;
#pragma unroll
        for (int d = 0; d < DPL; ++d) q[r][d] = qp[d];
    }
    const int lo = npad;
    const int hi = (t0 + R - 1 < p.T ? t0 + R - 1 : p.T - 1);   // the last row's last key; row r stops at key t0 + r
    const float* kb = base + p.k_off + kvh * HD + li * DPL;
    const float* vb = base + p.v_off + kvh * HD + li * DPL;
    float m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) m[r] = -INFINITY;
    for (int s0 = lo + g; s0 <= hi; s0 += 4 * UN) {          // pass 1: row maxima
        float kr[UN][DPL];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int sc = s0 + 4 * u <= hi ? s0 + 4 * u : hi;
            const float* kp = kb + (size_t)sc * p.ld;
#pragma unroll
            for (int e = 0; e < DPL; ++e) kr[u][e] = kp[e];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (s0 + 4 * u > t0 + r) continue;               // (wave-uniform within a 16-lane group: all its lanes hold the same key)
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < DPL; ++e) d += q[r][e] * kr[u][e];
                d = group16_sum(d) * scale;
                m[r] = fmaxf(m[r], d);
            }
    }
    float l[R], acc[R][DPL];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m[r] = fmaxf(m[r], __shfl_xor(m[r], 16));
        m[r] = fmaxf(m[r], __shfl_xor(m[r], 32));
        l[r] = 0.f;
#pragma unroll
        for (int e = 0; e < DPL; ++e) acc[r][e] = 0.f;
    }
    for (int s0 = lo + g; s0 <= hi; s0 += 4 * UN) {          // pass 2: exp, sum, PV
        float kr[UN][DPL], vr[UN][DPL];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int sc = s0 + 4 * u <= hi ? s0 + 4 * u : hi;
            const float* kp = kb + (size_t)sc * p.ld;
            const float* vp = vb + (size_t)sc * p.ld;
#pragma unroll
            for (int e = 0; e < DPL; ++e) { kr[u][e] = kp[e]; vr[u][e] = vp[e]; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (s0 + 4 * u > t0 + r) continue;
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < DPL; ++e) d += q[r][e] * kr[u][e];
                d = group16_sum(d) * scale;
                const float pr = expf(d - m[r]);
                l[r] += pr;
#pragma unroll
                for (int e = 0; e < DPL; ++e) acc[r][e] += pr * vr[u][e];
            }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        l[r] += __shfl_xor(l[r], 16);
        l[r] += __shfl_xor(l[r], 32);
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            acc[r][e] += __shfl_xor(acc[r][e], 16);
            acc[r][e] += __shfl_xor(acc[r][e], 32);
        }
        if (g != 0 || t0 + r >= p.T) continue;
        const size_t ooff = ((size_t)b * p.T + t0 + r) * p.ldo + h * HD + li * DPL;
        float* orow = p.out + ooff;
        bf16_t* orow16 = reinterpret_cast<bf16_t*>(p.out16) + ooff;
        const float inv = 1.f / l[r];
#pragma unroll
        for (int e = 0; e < DPL; ++e) {
            const float v = live[r] ? acc[r][e] * inv : 0.f;   // (a left-pad query row: zeros, never read downstream)
            if (p.out16) orow16[e] = f32_to_bf16(v); else orow[e] = v;
        }
    }
}

void launch_attn_rows(const AttnRowsParams& p, hipStream_t st) {
    // full-causal attention over more than a few rows (the talker's prefill): four query rows per wave (QTTS_ATTN_ROWS4=0: one)
    if (p.window <= 0 && p.T >= 8 && (p.hd == 64 || p.hd == 128) && QTTS_OPT_ON("QTTS_ATTN_ROWS4")) {
        const int64_t items = (int64_t)p.B * p.nh * ((p.T + 3) / 4);
        const int grid4 = (int)((items + 3) / 4);
        if (p.hd == 64) hipLaunchKernelGGL(attn_rows4_kernel<64>, dim3(grid4), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(attn_rows4_kernel<128>, dim3(grid4), dim3(256), 0, st, p);
        QTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    const int64_t total = (int64_t)p.B * p.nh * p.T;
    const int grid = (int)((total + 3) / 4);
    if (p.hd == 64) hipLaunchKernelGGL(attn_rows_kernel<64>, dim3(grid), dim3(256), 0, st, p);
    else if (p.hd == 128) hipLaunchKernelGGL(attn_rows_kernel<128>, dim3(grid), dim3(256), 0, st, p);
    else throw Error(QTTS_ERR_ARG, "attn_rows: head_dim must be 64 or 128");
    QTTS_CHECK_HIP(hipGetLastError());
}

// =================================================================================== KV cache helpers (attn_helpers.h)
// element offset of (layer, sequence b, position s, kv head) in a pool [layer][page][kvh][16][hd]
__device__ inline size_t kv_offset(const KvCache& c, int layer, int b, int s, int kvh) {
    const int page = c.page_table[b * c.pages_per_seq + (s >> 4)];
    return ((((size_t)layer * c.n_pages + page) * c.nkv + kvh) * 16 + (s & 15)) * c.hd;
}

// =================================================================================== qknorm_rope_store
// q and k heads: norm + RoPE in place, k also to the cache; v heads are copied to the cache.
// Round 6: a workgroup per (b, t) row, its four waves dealing the row's heads -- the rotation's cosine / sine depend on the position and the lane only, so a
// wave computes them ONCE for its 6-8 heads (one wave per (b, t, head) spent its time in cosf / sinf: 23.7 us for the 1 536-row prefill of config 4, VALU-bound;
// the same statements per element, so the results are bit-identical).
template <typename KVT>
__global__ __launch_bounds__(256) void qknorm_rope_store_kernel(QkNormRopeParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int heads = p.nh + 2 * p.nkv;
    const int t = (int)(blockIdx.x % p.T);
    const int b = (int)(blockIdx.x / p.T);
    const int npad = p.n_pad[b];
    if (t < npad) return;  // pad rows: K/V slots stay unused (masked by s < n_pad everywhere)
    const int hd = p.hd, half = hd / 2;
    KVT* kc = reinterpret_cast<KVT*>(p.kv.k);
    KVT* vc = reinterpret_cast<KVT*>(p.kv.v);
    const float pos = (float)(t - npad);
    const float ang = pos * p.inv_freq[hd == 128 ? lane : (lane & 31)];
    const float c = cosf(ang), s = sinf(ang);
    // the wave's heads (wave, wave + 4, ...) in groups of up to eight: all of a group's elements are requested before the first one is used (a head is a
    // load -> wave sum -> store chain; eight chains in flight instead of one after the other)
    float* row = p.qkv + ((size_t)b * p.T + t) * p.ld;
    for (int h0 = wave; h0 < heads; h0 += 32) {
        float xa[8], xb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int hh = h0 + 4 * i;
            const float* v = row + (hh < heads ? hh : h0) * hd;
            xa[i] = lane < hd ? v[lane] : 0.f;
            xb[i] = lane + 64 < hd ? v[lane + 64] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int hh = h0 + 4 * i;
            if (hh >= heads) break;
            float* v = row + hh * hd;
            if (hh >= p.nh + p.nkv) {  // value head: straight copy (or, for a transposed-V cache, dim-major inside the page)
                const int kvh = hh - p.nh - p.nkv;
                const size_t o = kv_offset(p.kv, p.layer, b, t, kvh);
                if (p.kv.vt) {
                    const size_t pg = o - (size_t)(t & 15) * hd;              // start of this (page, kv head) block
                    if (lane < hd) vc[pg + (size_t)lane * 16 + (t & 15)] = kv_cast<KVT>(xa[i]);
                    if (lane + 64 < hd) vc[pg + (size_t)(lane + 64) * 16 + (t & 15)] = kv_cast<KVT>(xb[i]);
                } else {
                    if (lane < hd) vc[o + lane] = kv_cast<KVT>(xa[i]);
                    if (lane + 64 < hd) vc[o + lane + 64] = kv_cast<KVT>(xb[i]);
                }
                continue;
            }
            const bool is_k = hh >= p.nh;
            const float* w = is_k ? p.kw : p.qw;
            // hd <= 128: lane owns d = lane and d + 64 (second only when hd == 128) -> pairs (d, d+half)
            float x0 = xa[i], x1 = xb[i];
            float ss = wave_sum64(x0 * x0 + x1 * x1);
            const float r = rsqrtf(ss / (float)hd + p.eps);
            x0 = w[lane < hd ? lane : 0] * (x0 * r);
            x1 = (lane + 64 < hd) ? w[lane + 64] * (x1 * r) : 0.f;
            float o0, o1;
            if (hd == 128) {  // pair (lane, lane+64)
                o0 = x0 * c - x1 * s;
                o1 = x1 * c + x0 * s;
            } else {          // hd == 64: pair (lane, lane^32) inside the wave
                const float other = __shfl_xor(x0, 32);
                o0 = (lane < 32) ? x0 * c - other * s : x0 * c + other * s;
                o1 = 0.f;
            }
            if (lane < hd) v[lane] = o0;
            if (lane + 64 < hd) v[lane + 64] = o1;
            if (is_k) {
                const size_t o = kv_offset(p.kv, p.layer, b, t, hh - p.nh);
                if (lane < hd) kc[o + lane] = kv_cast<KVT>(o0);
                if (lane + 64 < hd) kc[o + lane + 64] = kv_cast<KVT>(o1);
            }
        }
    }
    (void)half;
}

void launch_qknorm_rope_store(const QkNormRopeParams& p, hipStream_t st) {
    QTTS_REQUIRE(p.hd == 64 || p.hd == 128, QTTS_ERR_ARG, "qknorm_rope_store: head_dim 64|128");
    const int grid = p.B * p.T;                        // a workgroup per (b, t) row
    if (p.kv.bf16) hipLaunchKernelGGL(qknorm_rope_store_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(qknorm_rope_store_kernel<float>, dim3(grid), dim3(256), 0, st, p);
    QTTS_CHECK_HIP(hipGetLastError());
}

// =================================================================================== attn_decode
// grid = B * nkv workgroups of 256 threads.  HD = 128 only (talker and code predictor).
// Latency-first structure (this kernel runs 103x per frame on 17..300 keys): all loads that do not depend on
// the freshly produced qkv row -- the first KV chunk of every 16-lane key group, the length and pad scalars --
// are issued at kernel entry; q/k RMSNorm + RoPE of the new tokens runs underneath them; later KV chunks are
// prefetched one chunk ahead.  Key s is owned by lane group (s % 16); new keys come from LDS, old ones from
// the paged cache (page = table[b][s/16], or b*pages_per_seq + s/16 when the pool is laid out contiguously).
// The per-key inner loops are branch-free over the 4 keys of a chunk (independent dot products interleave), use DPP
// row reductions, and handle the 1-2 NEW keys in a separate tiny pass; the 16 key groups combine through LDS.
// LDS: qs[NQ][128] | kn[n_new][128] | vn[n_new][128] | red[16][NQ][128] | sc[NQ][max_len] | stat[NQ]
#ifndef QTTS_ATTN_TAIL_BATCH
#define QTTS_ATTN_TAIL_BATCH 0
#endif
// CT (all decode kernels): the pool is laid out contiguously (page = b * pages_per_seq + s / 16 -- what the engine allocates): a
// compile-time fact, because with `CT ? ... : page_table[...]` decided at run time every K / V request sat behind a
// control-flow join at which the compiler waits for ALL outstanding loads (`s_waitcnt vmcnt(0)`: one side of the join has a
// page-table load pending) -- the speculative chunk requests of attn_tk went out ONE AT A TIME, a memory round trip each
// (round-2 in-kernel timestamps: 5.1 of its 9.5 us; found in the ISA, profiles/r02_tstamp_frame.md).
template <typename KVT, int NQ, bool CT>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecodeParams p) {
    constexpr int HD = 128;
    constexpr int CH = 4;                      // keys per lane group per chunk (64 keys per workgroup chunk)
    constexpr int KW = sizeof(KVT) == 2 ? 1 : 2;   // 16-B vectors per key per lane (8 dims)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float sm_ad[];
    const int GQ = p.nh / p.nkv;
    float* qs = sm_ad;                         // [NQ][HD]   (query index = t*GQ + gq)
    float* kn = qs + NQ * HD;                  // [n_new][HD]
    float* vn = kn + p.n_new * HD;             // [n_new][HD]
    float* red = vn + p.n_new * HD;            // [16][NQ][HD]
    float* sc = red + 16 * NQ * HD;            // [NQ][max_len]
    float* stat = sc + NQ * p.max_len;         // [NQ]

    const int b = blockIdx.x / p.nkv, kvh = blockIdx.x % p.nkv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = tid >> 4, li = tid & 15;
    const KVT* kc = reinterpret_cast<const KVT*>(p.kv.k);
    const KVT* vc = reinterpret_cast<const KVT*>(p.kv.v);
    const int pool_keys = p.kv.pages_per_seq * 16;

    auto kv_off = [&](int s) -> size_t {       // element offset of key s, this lane's 8 dims
        s = s < pool_keys ? s : pool_keys - 1;  // speculative loads stay inside the sequence's pages
        const int page = CT ? b * p.kv.pages_per_seq + (s >> 4) : p.kv.page_table[b * p.kv.pages_per_seq + (s >> 4)];
        return ((((size_t)p.layer * p.kv.n_pages + page) * p.kv.nkv + kvh) * 16 + (s & 15)) * HD + li * 8;
    };
    auto load_chunk = [&](u32x4 (&r)[CH][KW], const KVT* base, int c) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const u32x4* src = reinterpret_cast<const u32x4*>(base + kv_off(g + 16 * (c * CH + i)));
#pragma unroll
            for (int w = 0; w < KW; ++w) r[i][w] = src[w];
        }
    };
    auto unpack = [&](const u32x4 (&r)[KW], float (&x)[8]) {
        if constexpr (KW == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[2 * e] = __uint_as_float(r[0][e] << 16);
                x[2 * e + 1] = __uint_as_float(r[0][e] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = __uint_as_float(r[0][e]); x[4 + e] = __uint_as_float(r[1][e]); }
        }
    };

    // ---- 0. loads that do not depend on this step's qkv: the first K and V chunk speculatively, then -- as soon as
    // the length scalar is back -- every further chunk that is needed (up to NPRE chunks = 256 keys with a bf16
    // cache), so the whole KV read of a typical step is ONE latency round instead of one per chunk.
    constexpr int NPRE = KW == 1 ? 4 : 2;
    u32x4 kR[NPRE][CH][KW], vR[NPRE][CH][KW];
    load_chunk(kR[0], kc, 0);
    load_chunk(vR[0], vc, 0);
    const bool deep = p.max_len > 64;          // long-sequence stack (talker): the second chunk is almost always needed
    if (deep) { load_chunk(kR[1], kc, 1); load_chunk(vR[1], vc, 1); }
    // this step's q/k/v rows are requested BEFORE the length-dependent chunks: loads return in order, and stage 1
    // (norm + RoPE + append) should overlap the rest of the KV stream instead of queueing behind it
    const int nvec = NQ + 2 * p.n_new;  // q vectors, then k, then v
    float x0v[2], x1v[2];               // up to 2 vectors per wave (nvec <= 8)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int vi = wave + 4 * r;
        x0v[r] = x1v[r] = 0.f;
        if (vi < nvec) {
            int t, col;
            if (vi < NQ) { t = vi / GQ; col = (kvh * GQ + vi % GQ) * HD; }
            else if (vi < NQ + p.n_new) { t = vi - NQ; col = (p.nh + kvh) * HD; }
            else { t = vi - NQ - p.n_new; col = (p.nh + p.nkv + kvh) * HD; }
            const float* src = p.qkv + ((size_t)t * p.B + b) * p.ld + col;
            x0v[r] = src[lane]; x1v[r] = src[lane + 64];
        }
    }
    const int S0 = p.len_dev ? *p.len_dev : p.len_static;   // KV length before this step
    const int npad = p.n_pad ? p.n_pad[b] : 0;
    const int done = p.done_flag ? *p.done_flag : 0;
    const int S1 = S0 + p.n_new;               // total keys
    const int nchunk = (S1 + 16 * CH - 1) / (16 * CH);
#pragma unroll
    for (int c = 1; c < NPRE; ++c)
        if (c < nchunk && !(deep && c == 1)) { load_chunk(kR[c], kc, c); load_chunk(vR[c], vc, c); }

    // ---- 1. q/k RMSNorm + RoPE for the new tokens (one wave per vector), K/V append
    if (done) return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int vi = wave + 4 * r;
        if (vi >= nvec) continue;
        int t;
        const float* w = nullptr;
        float* dst;
        if (vi < NQ) { t = vi / GQ; w = p.qw; dst = qs + vi * HD; }
        else if (vi < NQ + p.n_new) { t = vi - NQ; w = p.kw; dst = kn + t * HD; }
        else { t = vi - NQ - p.n_new; dst = vn + t * HD; }
        float x0 = x0v[r], x1 = x1v[r];
        if (w) {
            const float ss = wave_sum64_dpp(x0 * x0 + x1 * x1);
            const float rs = rsqrtf(ss / (float)HD + p.eps);
            x0 = w[lane] * (x0 * rs);
            x1 = w[lane + 64] * (x1 * rs);
            const float ang = (float)(S0 + t - npad) * p.inv_freq[lane];
            const float c = cosf(ang), sn = sinf(ang);
            const float o0 = x0 * c - x1 * sn, o1 = x1 * c + x0 * sn;
            x0 = o0; x1 = o1;
        }
        if (vi >= NQ) {  // K or V of a new token: round through the cache type, append
            const int s = S0 + t;
            const int page = CT ? b * p.kv.pages_per_seq + (s >> 4) : p.kv.page_table[b * p.kv.pages_per_seq + (s >> 4)];
            const size_t o = ((((size_t)p.layer * p.kv.n_pages + page) * p.kv.nkv + kvh) * 16 + (s & 15)) * HD;
            KVT* cdst = reinterpret_cast<KVT*>(vi < NQ + p.n_new ? p.kv.k : p.kv.v);
            const KVT h0 = kv_cast<KVT>(x0), h1 = kv_cast<KVT>(x1);
            cdst[o + lane] = h0; cdst[o + lane + 64] = h1;
            x0 = kv_load(&h0); x1 = kv_load(&h1);
        }
        dst[lane] = x0; dst[lane + 64] = x1;
    }
    __syncthreads();

    // ---- 2. scores: 16 lanes per key (8 dims each), 16 keys per sweep.  Old keys (s < S0) come from the cache
    // registers, branch-free; the n_new fresh keys from LDS in their own pass.
    const float scale = rsqrtf((float)HD);
    float qreg[NQ][8];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int e = 0; e < 8; ++e) qreg[qi][e] = qs[qi * HD + li * 8 + e];
    auto score_chunk = [&](const u32x4 (&r)[CH][KW], int c) {
        float d[CH][NQ];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            float kx[8];
            unpack(r[i], kx);
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) a += qreg[qi][e] * kx[e];
                d[i][qi] = a;
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) d[i][qi] = row16_sum(d[i][qi]);
        if (li == 0) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int s = g + 16 * (c * CH + i);
                if (s < S0) {
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) sc[qi * p.max_len + s] = s >= npad ? d[i][qi] * scale : -INFINITY;
                }
            }
        }
    };
#pragma unroll
    for (int c = 0; c < NPRE; ++c)
        if (c < nchunk) score_chunk(kR[c], c);
#if QTTS_ATTN_TAIL_BATCH
    // A/B variant (build.py VARIANTS): beyond the prefetch window the K registers are free again, so the tail is read
    // NPRE chunks (256 keys bf16) per latency round instead of one -- what a > 20 s utterance runs every layer.
    for (int c0 = NPRE; c0 < nchunk; c0 += NPRE) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
            if (c0 + j < nchunk) load_chunk(kR[j], kc, c0 + j);
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
            if (c0 + j < nchunk) score_chunk(kR[j], c0 + j);
    }
#else
    for (int c = NPRE; c < nchunk; ++c) {       // very long sequences: plain loop
        u32x4 kB[CH][KW];
        load_chunk(kB, kc, c);
        score_chunk(kB, c);
    }
#endif
    for (int t = 0; t < p.n_new; ++t) {         // the fresh keys: key S0+t belongs to group (S0+t) % 16
        const int s = S0 + t;
        if (g == (s & 15)) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) a += qreg[qi][e] * kn[t * HD + li * 8 + e];
                a = row16_sum(a) * scale;
                if (li == 0) sc[qi * p.max_len + s] = (s >= npad && s <= S0 + qi / GQ) ? a : -INFINITY;
            }
        }
    }
    __syncthreads();
    // ---- 3. softmax statistics: wave qi handles query qi
    if (wave < NQ) {
        const int qi = wave;
        float m = -INFINITY;
        for (int s = lane; s < S1; s += 64) m = fmaxf(m, sc[qi * p.max_len + s]);
        m = wave_max64_dpp(m);
        float l = 0.f;
        for (int s = lane; s < S1; s += 64) {
            const float e = expf(sc[qi * p.max_len + s] - m);
            sc[qi * p.max_len + s] = e;
            l += e;
        }
        l = wave_sum64_dpp(l);
        if (lane == 0) stat[qi] = 1.f / l;
    }
    __syncthreads();
    // ---- 4. PV: same key ownership, accumulate 8 dims per lane per query
    float acc[NQ][8];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[qi][e] = 0.f;
    auto pv_chunk = [&](const u32x4 (&r)[CH][KW], int c) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int s = g + 16 * (c * CH + i);
            const bool valid = s < S0 && s >= npad;      // left-pad slots were never written; slots >= S0 hold nothing yet
            float vx[8];
            unpack(r[i], vx);
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const float pr = valid ? sc[qi * p.max_len + s] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[qi][e] += pr * (valid ? vx[e] : 0.f);
            }
        }
    };
#pragma unroll
    for (int c = 0; c < NPRE; ++c)
        if (c < nchunk) pv_chunk(vR[c], c);
#if QTTS_ATTN_TAIL_BATCH
    for (int c0 = NPRE; c0 < nchunk; c0 += NPRE) {
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
            if (c0 + j < nchunk) load_chunk(vR[j], vc, c0 + j);
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
            if (c0 + j < nchunk) pv_chunk(vR[j], c0 + j);
    }
#else
    for (int c = NPRE; c < nchunk; ++c) {
        u32x4 vB[CH][KW];
        load_chunk(vB, vc, c);
        pv_chunk(vB, c);
    }
#endif
    for (int t = 0; t < p.n_new; ++t) {
        const int s = S0 + t;
        if (g == (s & 15) && s >= npad) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const float pr = sc[qi * p.max_len + s];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[qi][e] += pr * vn[t * HD + li * 8 + e];
            }
        }
    }
    // the 16 key groups combine through LDS in a fixed order
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        float4* dst = reinterpret_cast<float4*>(red + (g * NQ + qi) * HD + li * 8);
        dst[0] = make_float4(acc[qi][0], acc[qi][1], acc[qi][2], acc[qi][3]);
        dst[1] = make_float4(acc[qi][4], acc[qi][5], acc[qi][6], acc[qi][7]);
    }
    __syncthreads();
    for (int i = tid; i < NQ * HD; i += 256) {
        const int qi = i / HD, d = i % HD;
        float v = 0.f;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) v += red[(gg * NQ + qi) * HD + d];
        const int t = qi / GQ, gq = qi % GQ;
        const size_t o = ((size_t)t * p.B + b) * p.ldo + (kvh * GQ + gq) * HD + d;
        if (p.out_bf16) reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(v * stat[qi]);
        else p.out[o] = v * stat[qi];
    }
}

// =================================================================================== attn_cp (promoted in round 2)
// The code predictor's single-token passes (70 of the 103 attention launches of a frame) attend over at most 16 keys, yet
// go through the general kernel above: 64-key speculative chunks, five workgroup barriers, 16 key groups combined through
// LDS.  This kernel does the same arithmetic for exactly that case -- one new token, a static cache length S0 <= 15, no
// left padding, 1 or 2 query heads per kv head, head_dim 128 -- with one barrier:
//   stage 1 (4 waves):  q head 0 | q head 1 | k | v of the new token: RMSNorm + RoPE (q, k), round through the cache type and
//                       append (k, v), results to LDS.  This step's row, the norm weights and the cos | sin row of the (launch-time
//                       known) position are requested first, then the old K rows (key-major: lane = key * 4 + quarter, 32
//                       dims each) and the old V rows (lane owns dims 2 lane, 2 lane + 1 of every key: one 256-B row per
//                       request) of all 16 key slots, straight-line and unconditional (slots >= S0 re-read key 0).
//   stage 2 (wave = query): 32-dim partial dot + quad reduction, fp32 softmax across the 16 key slots (DPP), then
//                       out[d] = sum_k e_k * v[k][d] with e_k broadcast from its lane -- no cross-lane reduction for PV.
template <typename KVT, bool CT>
__global__ __launch_bounds__(256) void attn_cp_kernel(AttnDecodeParams p) {
    constexpr int HD = 128, MAXK = 16;
    constexpr int KW = sizeof(KVT) == 2 ? 4 : 8;          // 16-B vectors per 32-dim quarter of a key
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float qs[2][HD];
    __shared__ __attribute__((aligned(16))) float kn[HD];
    __shared__ float vn[HD];
    const int GQ = p.nh / p.nkv;
    QTTS_TS_BEGIN();
    const int b = blockIdx.x / p.nkv, kvh = blockIdx.x % p.nkv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kk = lane >> 2, qq = lane & 3;
    const int S0 = p.len_static, S1 = S0 + 1;
    const KVT* kc = reinterpret_cast<const KVT*>(p.kv.k);
    const KVT* vc = reinterpret_cast<const KVT*>(p.kv.v);
    auto key_base = [&](int s) -> size_t {                // element offset of key s (dim 0) in this sequence's pages
        const int page = CT ? b * p.kv.pages_per_seq + (s >> 4) : p.kv.page_table[b * p.kv.pages_per_seq + (s >> 4)];
        return ((((size_t)p.layer * p.kv.n_pages + page) * p.kv.nkv + kvh) * 16 + (s & 15)) * HD;
    };
    // ---- 0. this step's row FIRST (loads return in request order: the norm / RoPE stage then runs while the cache rows are still
    // on their way): wave 0 / 1 -> q heads, wave 2 -> k, wave 3 -> v (with GQ == 1 wave 1 has nothing to do)
    const bool has_vec = wave >= 2 || wave < GQ;
    float x0 = 0.f, x1 = 0.f;
    if (has_vec) {
        const int col = wave < 2 ? (kvh * GQ + wave) * HD : (wave == 2 ? (p.nh + kvh) * HD : (p.nh + p.nkv + kvh) * HD);
        const float* src = p.qkv + (size_t)b * p.ld + col;
        x0 = src[lane]; x1 = src[lane + 64];
    }
    // (norm weights and RoPE frequencies requested here, unconditionally: behind the early-exit test they would cost stage 1 a
    // second memory round trip)
    const float* nw = wave == 2 ? p.kw : p.qw;
    const float nw0 = nw[lane], nw1 = nw[lane + 64], invf = p.inv_freq[lane];
    const bool rtab = p.rope_cs && S0 < p.rope_cs_n;                  // (kernel-uniform; the row is requested either way)
    const float* rrow = rtab ? p.rope_cs + (size_t)S0 * 128 : p.inv_freq;
    const float ctab = rrow[lane], stab = rrow[rtab ? 64 + lane : lane];
    // cache reads (do not depend on this step's qkv row); only the query waves need them.  Straight-line and unconditional
    // inside the (wave-uniform) branch: key slots >= S0 re-read key 0 and are dropped at the point of use -- a conditional load
    // is merged with the register's previous value, and the compiler waited for each of the 32 V loads before issuing the next
    // (round 2: the kernel's time grew with the pass number, 2.5 us to first data on average).  V: lane <- dims 2 lane, 2 lane + 1
    // of every key (one 256-B row per request instead of two 128-B halves).
    struct alignas(2 * sizeof(KVT)) VPair { KVT a, b; };
    u32x4 kr[KW];
    VPair vr[MAXK];
    if (wave < GQ) {
        const u32x4* ksrc = reinterpret_cast<const u32x4*>(kc + key_base(kk < S0 ? kk : 0) + qq * 32);
#pragma unroll
        for (int w = 0; w < KW; ++w) kr[w] = ksrc[w];
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            vr[k] = *reinterpret_cast<const VPair*>(vc + key_base(k < S0 ? k : 0) + 2 * lane);
    }
    const int done = p.done_flag ? *p.done_flag : 0;
    QTTS_TS(1);
    if (done) return;
    QTTS_TS_DRAINED(2);                    // cache rows and this step's qkv row have arrived
    // ---- 1. q/k RMSNorm + RoPE at position S0, K/V append
    if (has_vec) {
        if (wave <= 2) {
            const float ss = wave_sum64_dpp(x0 * x0 + x1 * x1);
            const float rs = rsqrtf(ss / (float)HD + p.eps);
            x0 = nw0 * (x0 * rs);
            x1 = nw1 * (x1 * rs);
            float c = ctab, sn = stab;
            if (!rtab) { const float ang = (float)S0 * invf; c = cosf(ang); sn = sinf(ang); }
            const float o0 = x0 * c - x1 * sn, o1 = x1 * c + x0 * sn;
            x0 = o0; x1 = o1;
        }
        if (wave >= 2) {
            const size_t o = key_base(S0);
            KVT* cdst = reinterpret_cast<KVT*>(wave == 2 ? p.kv.k : p.kv.v);
            const KVT h0 = kv_cast<KVT>(x0), h1 = kv_cast<KVT>(x1);
            cdst[o + lane] = h0; cdst[o + lane + 64] = h1;
            x0 = kv_load(&h0); x1 = kv_load(&h1);
        }
        float* dst = wave < 2 ? qs[wave] : (wave == 2 ? kn : vn);
        dst[lane] = x0; dst[lane + 64] = x1;
    }
    QTTS_TS_DRAINED(3);                    // norm + RoPE + append done
    __syncthreads();
    QTTS_TS(4);
    if (wave >= GQ) return;
    // ---- 2. one wave per query head
    const float* q = qs[wave] + qq * 32;
    float kx[32];
    if (kk == S0) {
#pragma unroll
        for (int e = 0; e < 32; ++e) kx[e] = kn[qq * 32 + e];
    } else {
#pragma unroll
        for (int w = 0; w < KW; ++w) {
            if constexpr (KW == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kx[w * 8 + 2 * e] = __uint_as_float(kr[w][e] << 16);
                    kx[w * 8 + 2 * e + 1] = __uint_as_float(kr[w][e] & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) kx[w * 4 + e] = __uint_as_float(kr[w][e]);
            }
        }
    }
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
            // (v_readlane with a compile-time lane: a few cycles, where the __shfl this replaces was a ds_bpermute round trip each)
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
    const size_t o = (size_t)b * p.ldo + (kvh * GQ + wave) * HD + 2 * lane;      // this lane's two dims
    if (p.out_bf16) {
        const unsigned pk = (unsigned)f32_to_bf16(acc0 * inv) | ((unsigned)f32_to_bf16(acc1 * inv) << 16);
        *reinterpret_cast<unsigned*>(reinterpret_cast<bf16_t*>(p.out) + o) = pk;
    } else { p.out[o] = acc0 * inv; p.out[o + 1] = acc1 * inv; }
    QTTS_TS_DRAINED(5);
    QTTS_TS_END(attn, 1, S0, 0);
}

// Pass 0 of the code predictor: an empty cache and TWO new tokens [past_hidden, embedding of codebook 0] (M:1671-1680).
// Token 0 attends to itself only, so its output IS its v row (softmax of one score = 1, exactly as the general kernel computes
// it); token 1 attends to both.  One workgroup per (sequence, kv head): 8 vectors (q of 2 tokens x up to 2 heads, k and v of
// both tokens) get RMSNorm + RoPE at positions 0 / 1 two per wave, k / v are appended, then wave gq < GQ forms the two-key
// softmax of query (token 1, head gq).
template <typename KVT, bool CT>
__global__ __launch_bounds__(256) void attn_cp0_kernel(AttnDecodeParams p) {
    constexpr int HD = 128;
    __shared__ float q1[2][HD];            // normed + roped q of token 1, per head
    __shared__ float kn[2][HD], vn[2][HD]; // k (normed, roped, rounded through the cache type) and v (rounded) of both tokens
    const int GQ = p.nh / p.nkv;
    const int b = blockIdx.x / p.nkv, kvh = blockIdx.x % p.nkv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // vector vi = wave + 4 * r:  0..3 -> q(token vi / 2, head vi % 2);  4, 5 -> k(token vi - 4);  6, 7 -> v(token vi - 6)
    float x0v[2], x1v[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int vi = wave + 4 * r;
        x0v[r] = x1v[r] = 0.f;
        int t, col;
        bool have = true;
        if (vi < 4) { t = vi >> 1; col = (kvh * GQ + (vi & 1)) * HD; have = (vi & 1) < GQ && t == 1; }   // token 0's q is never used
        else if (vi < 6) { t = vi - 4; col = (p.nh + kvh) * HD; }
        else { t = vi - 6; col = (p.nh + p.nkv + kvh) * HD; }
        if (have) {
            const float* src = p.qkv + ((size_t)t * p.B + b) * p.ld + col;
            x0v[r] = src[lane]; x1v[r] = src[lane + 64];
        }
    }
    const int done = p.done_flag ? *p.done_flag : 0;
    if (done) return;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int vi = wave + 4 * r;
        const int t = vi < 4 ? (vi >> 1) : (vi < 6 ? vi - 4 : vi - 6);
        if (vi < 4 && ((vi & 1) >= GQ || t == 0)) continue;
        const float* w = vi < 4 ? p.qw : (vi < 6 ? p.kw : nullptr);
        float x0 = x0v[r], x1 = x1v[r];
        if (w) {
            const float ss = wave_sum64_dpp(x0 * x0 + x1 * x1);
            const float rs = rsqrtf(ss / (float)HD + p.eps);
            x0 = w[lane] * (x0 * rs);
            x1 = w[lane + 64] * (x1 * rs);
            const float ang = (float)t * p.inv_freq[lane];
            const float c = cosf(ang), sn = sinf(ang);
            const float o0 = x0 * c - x1 * sn, o1 = x1 * c + x0 * sn;
            x0 = o0; x1 = o1;
        }
        if (vi >= 4) {                      // K or V of token t: round through the cache type, append at slot t of page 0
            const int page = CT ? b * p.kv.pages_per_seq : p.kv.page_table[b * p.kv.pages_per_seq];
            const size_t o = ((((size_t)p.layer * p.kv.n_pages + page) * p.kv.nkv + kvh) * 16 + t) * HD;
            KVT* cdst = reinterpret_cast<KVT*>(vi < 6 ? p.kv.k : p.kv.v);
            const KVT h0 = kv_cast<KVT>(x0), h1 = kv_cast<KVT>(x1);
            cdst[o + lane] = h0; cdst[o + lane + 64] = h1;
            x0 = kv_load(&h0); x1 = kv_load(&h1);
            float* dst = vi < 6 ? kn[t] : vn[t];
            dst[lane] = x0; dst[lane + 64] = x1;
            if (vi >= 6 && t == 0) {        // token 0's attention output = its own v row, for every head of this kv group
                for (int gq = 0; gq < GQ; ++gq) {
                    const size_t oo = (size_t)b * p.ldo + (kvh * GQ + gq) * HD;
                    if (p.out_bf16) {
                        reinterpret_cast<bf16_t*>(p.out)[oo + lane] = f32_to_bf16(x0);
                        reinterpret_cast<bf16_t*>(p.out)[oo + lane + 64] = f32_to_bf16(x1);
                    } else { p.out[oo + lane] = x0; p.out[oo + lane + 64] = x1; }
                }
            }
        } else if (t == 1) {
            q1[vi & 1][lane] = x0; q1[vi & 1][lane + 64] = x1;
        }
    }
    __syncthreads();
    if (wave >= GQ) return;
    // query (token 1, head `wave`) over keys 0 and 1
    const float qa = q1[wave][lane], qb = q1[wave][lane + 64];
    const float scale = rsqrtf((float)HD);
    const float s0 = wave_sum64_dpp(qa * kn[0][lane] + qb * kn[0][lane + 64]) * scale;
    const float s1 = wave_sum64_dpp(qa * kn[1][lane] + qb * kn[1][lane + 64]) * scale;
    const float m = fmaxf(s0, s1);
    const float e0 = expf(s0 - m), e1 = expf(s1 - m);
    const float inv = 1.f / (e0 + e1);
    const float o0 = (e0 * vn[0][lane] + e1 * vn[1][lane]) * inv;
    const float o1 = (e0 * vn[0][lane + 64] + e1 * vn[1][lane + 64]) * inv;
    const size_t oo = ((size_t)p.B + b) * p.ldo + (kvh * GQ + wave) * HD;     // row t * B + b with t = 1
    if (p.out_bf16) {
        reinterpret_cast<bf16_t*>(p.out)[oo + lane] = f32_to_bf16(o0);
        reinterpret_cast<bf16_t*>(p.out)[oo + lane + 64] = f32_to_bf16(o1);
    } else { p.out[oo + lane] = o0; p.out[oo + lane + 64] = o1; }
}


// =================================================================================== attn_tk (round 2)
// The talker's single-token decode attention (28 launches per frame; 12.7 us each in the general kernel above, ~60 us at 800
// keys; 6.5 us after the second half of round 2: contiguous-pool page index as a template parameter CT, v_exp_f32).  Same loads, same key ownership (key s -> 16-lane group s % 16, 8 dims per lane, 64 keys per chunk, the first chunks
// requested speculatively at kernel entry), but flash-decoding INSIDE the workgroup:
//   * no workgroup-wide score buffer and no softmax phase: every 16-lane group keeps its own running (max, sum, PV accumulator)
//     over its keys in registers (online softmax: a chunk of 4 keys costs one rescale); the 16 partial results are merged once,
//     out = sum_g acc_g e^(m_g - m) / sum_g l_g e^(m_g - m), in a fixed order;
//   * no staging barrier for the new token either: every wave norms / ropes q (and k, v) of the new token itself and
//     re-distributes it through a wave-private LDS slice (wave-level ordering only); wave 0 appends K / V to the cache;
//   * ONE workgroup barrier (in front of the merge) instead of four;
//   * beyond the register window (256 keys bf16 / 128 fp32) K and V chunks are read TOGETHER, NPRE chunks per latency round,
//     and folded into the running statistics -- a long utterance costs one round trip per 256 keys, not two per 64.
template <typename KVT, int GQ, bool CT>
__global__ __launch_bounds__(256) void attn_tk_kernel(AttnDecodeParams p) {
    constexpr int HD = 128, CH = 4;
    constexpr int KW = sizeof(KVT) == 2 ? 1 : 2;          // 16-B vectors per key per lane (8 dims)
    constexpr int NPRE = KW == 1 ? 4 : 2;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float xw[4][GQ + 2][HD];      // per wave: q heads, k, v of the new token
    __shared__ __attribute__((aligned(16))) float red[16][GQ][HD];
    __shared__ float gm[16][GQ], gl[16][GQ];

    QTTS_TS_BEGIN();
    const int b = blockIdx.x / p.nkv, kvh = blockIdx.x % p.nkv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = tid >> 4, li = tid & 15;
    const KVT* kc = reinterpret_cast<const KVT*>(p.kv.k);
    const KVT* vc = reinterpret_cast<const KVT*>(p.kv.v);
    const int pool_keys = p.kv.pages_per_seq * 16;

    auto kv_off = [&](int s) -> size_t {       // element offset of key s, this lane's 8 dims
        s = s < pool_keys ? s : pool_keys - 1;  // speculative loads stay inside the sequence's pages
        const int page = CT ? b * p.kv.pages_per_seq + (s >> 4) : p.kv.page_table[b * p.kv.pages_per_seq + (s >> 4)];
        return ((((size_t)p.layer * p.kv.n_pages + page) * p.kv.nkv + kvh) * 16 + (s & 15)) * HD + li * 8;
    };
    auto load_chunk = [&](u32x4 (&r)[CH][KW], const KVT* base, int c) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const u32x4* src = reinterpret_cast<const u32x4*>(base + kv_off(g + 16 * (c * CH + i)));
#pragma unroll
            for (int w = 0; w < KW; ++w) r[i][w] = src[w];
        }
    };
    auto unpack = [&](const u32x4 (&r)[KW], float (&x)[8]) {
        if constexpr (KW == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[2 * e] = __uint_as_float(r[0][e] << 16);
                x[2 * e + 1] = __uint_as_float(r[0][e] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = __uint_as_float(r[0][e]); x[4 + e] = __uint_as_float(r[1][e]); }
        }
    };

    // split-KV: workgroup y of gridDim.y handles the chunks [c0s, cend_static) of the (static) key capacity
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int cps = ((p.max_len + 16 * CH - 1) / (16 * CH) + nsplit - 1) / nsplit;     // chunks per split
    const int c0s = split * cps, cend_static = c0s + cps;
    // ---- 0. this step's row FIRST (loads return in request order: norm / RoPE of the new token then overlap the K / V stream):
    // EVERY wave fetches the GQ query heads, k and v of the new token (lane <- dims lane, lane + 64) and the norm weights
    float x0v[GQ + 2], x1v[GQ + 2];
#pragma unroll
    for (int vi = 0; vi < GQ + 2; ++vi) {
        const int col = vi < GQ ? (kvh * GQ + vi) * HD : (vi == GQ ? (p.nh + kvh) * HD : (p.nh + p.nkv + kvh) * HD);
        const float* src = p.qkv + (size_t)b * p.ld + col;
        x0v[vi] = src[lane]; x1v[vi] = src[lane + 64];
    }
    const float wq0 = p.qw[lane], wq1 = p.qw[lane + 64], wk0 = p.kw[lane], wk1 = p.kw[lane + 64];
    const float invf = p.inv_freq[lane];
    // loads that do not depend on this step's qkv row or on the length: the first K and V chunk(s)
    u32x4 kR[NPRE][CH][KW], vR[NPRE][CH][KW];
    load_chunk(kR[0], kc, c0s);
    load_chunk(vR[0], vc, c0s);
    const bool deep = p.max_len > 64 && cps > 1;
    if (deep) { load_chunk(kR[1], kc, c0s + 1); load_chunk(vR[1], vc, c0s + 1); }
    const int S0 = p.len_dev ? *p.len_dev : p.len_static;   // KV length before this step
    const int npad = p.n_pad ? p.n_pad[b] : 0;
    const int done = p.done_flag ? *p.done_flag : 0;
    const int S1 = S0 + 1;                                  // total keys
    const int nchunk = min(cend_static, (S0 + 16 * CH - 1) / (16 * CH));   // chunks of CACHED keys of this split (the new key comes from LDS)
#pragma unroll
    for (int c = 1; c < NPRE; ++c)
        if (c0s + c < nchunk && !(deep && c == 1)) { load_chunk(kR[c], kc, c0s + c); load_chunk(vR[c], vc, c0s + c); }
    if (done) return;
    QTTS_TS_DRAINED(1);                    // (tstamp build: phases 1..5 = arrived | normed | keys folded | barrier | stored)

    // ---- 1. q / k RMSNorm + RoPE of the new token, per wave; K / V append by wave 0 (rounded through the cache type: every
    // wave uses the rounded values, exactly what a later step will read back)
    const float ang = (float)(S0 - npad) * invf;
    const float cs = cosf(ang), sn = sinf(ang);
#pragma unroll
    for (int vi = 0; vi < GQ + 2; ++vi) {
        float x0 = x0v[vi], x1 = x1v[vi];
        if (vi <= GQ) {                                      // q heads and k
            const float ss = wave_sum64_dpp(x0 * x0 + x1 * x1);
            const float rs = rsqrtf(ss / (float)HD + p.eps);
            x0 = (vi < GQ ? wq0 : wk0) * (x0 * rs);
            x1 = (vi < GQ ? wq1 : wk1) * (x1 * rs);
            const float o0 = x0 * cs - x1 * sn, o1 = x1 * cs + x0 * sn;
            x0 = o0; x1 = o1;
        }
        if (vi >= GQ) {
            const KVT h0 = kv_cast<KVT>(x0), h1 = kv_cast<KVT>(x1);
            if (wave == 0 && split == 0) {
                const int page = CT ? b * p.kv.pages_per_seq + (S0 >> 4) : p.kv.page_table[b * p.kv.pages_per_seq + (S0 >> 4)];
                const size_t o = ((((size_t)p.layer * p.kv.n_pages + page) * p.kv.nkv + kvh) * 16 + (S0 & 15)) * HD;
                KVT* cdst = reinterpret_cast<KVT*>(vi == GQ ? p.kv.k : p.kv.v);
                cdst[o + lane] = h0; cdst[o + lane + 64] = h1;
            }
            x0 = kv_load(&h0); x1 = kv_load(&h1);
        }
        xw[wave][vi][lane] = x0; xw[wave][vi][lane + 64] = x1;
    }
    __builtin_amdgcn_wave_barrier();             // wave-private LDS slice: program order within the wave is all that is needed
    const float scale = rsqrtf((float)HD);
    float qreg[GQ][8];
#pragma unroll
    for (int qi = 0; qi < GQ; ++qi)
#pragma unroll
        for (int e = 0; e < 8; ++e) qreg[qi][e] = xw[wave][qi][li * 8 + e];
    QTTS_TS_DRAINED(2);

    // ---- 2. online softmax over this group's keys, everything in registers
    float m[GQ], l[GQ], acc[GQ][8];
#pragma unroll
    for (int qi = 0; qi < GQ; ++qi) {
        m[qi] = -INFINITY; l[qi] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[qi][e] = 0.f;
    }
    auto fold = [&](const float (&sc)[CH][GQ], const float (&vx)[CH][8]) {       // CH keys: scores (or -inf) and their V slices
#pragma unroll
        for (int qi = 0; qi < GQ; ++qi) {
            float mc = sc[0][qi];
#pragma unroll
            for (int i = 1; i < CH; ++i) mc = fmaxf(mc, sc[i][qi]);
            const float mn = fmaxf(m[qi], mc);
            if (mn == -INFINITY) continue;                   // nothing valid so far in this group
            const float f = att_exp<KVT>(m[qi] - mn);        // (m = -inf -> 0)
            float pr[CH], ps = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i) { pr[i] = att_exp<KVT>(sc[i][qi] - mn); ps += pr[i]; }
            l[qi] = l[qi] * f + ps;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = acc[qi][e] * f;
#pragma unroll
                for (int i = 0; i < CH; ++i) a += pr[i] * vx[i][e];
                acc[qi][e] = a;
            }
            m[qi] = mn;
        }
    };
    auto process = [&](const u32x4 (&kr)[CH][KW], const u32x4 (&vr)[CH][KW], int c) {
        float d[CH][GQ], vx[CH][8];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            float kx[8];
            unpack(kr[i], kx);
            unpack(vr[i], vx[i]);
#pragma unroll
            for (int qi = 0; qi < GQ; ++qi) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) a += qreg[qi][e] * kx[e];
                d[i][qi] = a;
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int s = g + 16 * (c * CH + i);
            const bool valid = s < S0 && s >= npad;          // left-pad slots were never written; slots >= S0 hold nothing yet
#pragma unroll
            for (int qi = 0; qi < GQ; ++qi) {
                const float r = row16_sum(d[i][qi]) * scale;     // (DPP row reduction: executed by every lane, then selected)
                d[i][qi] = valid ? r : -INFINITY;
            }
            if (!valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) vx[i][e] = 0.f;  // (0 * garbage must not become NaN)
            }
        }
        fold(d, vx);
    };
#pragma unroll
    for (int c = 0; c < NPRE; ++c)
        if (c0s + c < nchunk) process(kR[c], vR[c], c0s + c);
    for (int c0 = c0s + NPRE; c0 < nchunk; c0 += NPRE) {    // long sequences: NPRE chunks of K AND V per latency round
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
            if (c0 + j < nchunk) { load_chunk(kR[j], kc, c0 + j); load_chunk(vR[j], vc, c0 + j); }
#pragma unroll
        for (int j = 0; j < NPRE; ++j)
            if (c0 + j < nchunk) process(kR[j], vR[j], c0 + j);
    }
    const int split_new = min(nsplit - 1, (S0 / (16 * CH)) / cps);      // the split whose key range holds position S0
    if (split == split_new && g == (S0 & 15) && S0 >= npad) {   // the new key (position S0): k, v from this wave's LDS slice
        float d[CH][GQ], vx[CH][8];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
#pragma unroll
            for (int qi = 0; qi < GQ; ++qi) d[i][qi] = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e) vx[i][e] = 0.f;
        }
#pragma unroll
        for (int qi = 0; qi < GQ; ++qi) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a += qreg[qi][e] * xw[wave][GQ][li * 8 + e];
            d[0][qi] = row16_sum(a) * scale;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) vx[0][e] = xw[wave][GQ + 1][li * 8 + e];
        fold(d, vx);
    }
    (void)S1;
    // ---- 3. merge of the 16 groups (fixed order)
#pragma unroll
    for (int qi = 0; qi < GQ; ++qi) {
        float4* dst = reinterpret_cast<float4*>(&red[g][qi][li * 8]);
        dst[0] = make_float4(acc[qi][0], acc[qi][1], acc[qi][2], acc[qi][3]);
        dst[1] = make_float4(acc[qi][4], acc[qi][5], acc[qi][6], acc[qi][7]);
        if (li == 0) { gm[g][qi] = m[qi]; gl[g][qi] = l[qi]; }
    }
    QTTS_TS_DRAINED(3);
    __syncthreads();
    QTTS_TS(4);
    if (tid < GQ * HD) {
        const int qi = tid / HD, dd = tid % HD;
        float mm = gm[0][qi];
#pragma unroll
        for (int gg = 1; gg < 16; ++gg) mm = fmaxf(mm, gm[gg][qi]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) {
            const float f = gm[gg][qi] > -INFINITY ? att_exp<KVT>(gm[gg][qi] - mm) : 0.f;
            num += red[gg][qi][dd] * f;
            den += gl[gg][qi] * f;
        }
        if (nsplit > 1) {                                    // partial result of this split: numerator | max | denominator
            float* pp = p.part + (((size_t)blockIdx.x * nsplit + split) * GQ + qi) * (HD + 2);
            pp[dd] = num;
            if (dd == 0) { pp[HD] = mm; pp[HD + 1] = den; }
            QTTS_TS_DRAINED(5);
            QTTS_TS_END(attn, 2, S0, nsplit);
            return;
        }
        const size_t o = (size_t)b * p.ldo + (kvh * GQ + qi) * HD + dd;
        const float r = num / den;
        if (p.out_bf16) reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(r);
        else p.out[o] = r;
    }
    QTTS_TS_DRAINED(5);
    QTTS_TS_END(attn, 2, S0, nsplit);
}

// =================================================================================== attn_tk16 (round 3)
// The talker's single-token decode attention ON THE MATRIX PIPE (bf16 cache).  attn_tk above spends its key loop on the VALU:
// an 8-dim partial dot per lane and a 4-step DPP row reduction per key and query head (0.9 us per 64 keys per workgroup; 1.7 of
// its 5.3 us in-kernel at 130 keys, and what bounds a 60 s utterance).  Here both products are `v_mfma_f32_16x16x32_bf16`:
//   S = K q^T   A = 16 keys x 32 dims of a K page (row-major [key][dim]: a lane's 8 consecutive dims are one 16-B load),
//               B = q (query heads as columns; columns >= GQ are zero), 4 k-steps over the 128 dims;
//   O^T = V^T P A = 16 dims x 32 keys of the V pages, which this cache stores TRANSPOSED ([dim][16 keys], KvCache::vt): a
//               lane's 8 consecutive keys are one 16-B load as well -- no LDS transpose, no LDS at all before the final merge;
//               B = P (softmax numerators of 32 keys as bf16, heads as columns).
// The rows of the two S tiles of a 32-key block are PERMUTED keys: tile A row 4q + r = key 8q + r, tile B row 4q + r = key
// 8q + 4 + r, so that lane (head j, q) ends up with the scores of keys 8q .. 8q + 7 -- exactly the fragment the PV product wants
// from it as its B operand.  The softmax is the online one (running max / sum per head, accumulators rescaled per block), its
// statistics live in the lanes of a column (4 lanes per head, two xor-shuffles per block).  A workgroup is 4 waves, wave w
// takes the 32-key blocks w, w + 4, ... of its split; the new token's key (position S0) is folded in at the merge from this
// step's own row (fp32 q . k, as before).  Arithmetic: q, K, P, V enter the matrix pipe as bf16, accumulation fp32 -- the
// precision of the reference's own bf16 attention (M:634-657: bf16 q k^T, fp32 softmax cast to bf16, bf16 P V).
template <int GQ, bool CT>
__global__ __launch_bounds__(256) void attn_tk16_kernel(AttnDecodeParams p) {
    constexpr int HD = 128, NB = 2;            // NB: 32-key blocks per wave requested at kernel entry (4 waves x 2 x 32 = 256 keys)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float xw[4][GQ + 2][HD];      // per wave: q heads, k, v of the new token (fp32)
    __shared__ __attribute__((aligned(16))) float red[4][GQ][HD];          // per wave: un-normalised output
    __shared__ float gm[4][GQ], gl[4][GQ], snew[GQ];

    QTTS_TS_BEGIN();
    const int b = blockIdx.x / p.nkv, kvh = blockIdx.x % p.nkv;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const bf16_t* kc = reinterpret_cast<const bf16_t*>(p.kv.k);
    const bf16_t* vc = reinterpret_cast<const bf16_t*>(p.kv.v);
    const int pps = p.kv.pages_per_seq;

    // element offset of (page pg of this sequence, this kv head) -- the same for the K pool ([16][128]) and the V pool ([128][16])
    auto page_base = [&](int pg) -> size_t {
        // speculative READS stay inside the sequence's pages (what they fetch beyond the live length is dropped at use).  With a page
        // table (CT = false) every entry (b, pg < pps) must therefore be populated at create -- the engine reserves all pages of a
        // sequence up front.  The APPEND below never goes through the clamp: a step at capacity writes nothing (`fits`), it does not
        // overwrite the last page (the host refuses such a step first: max_seq check in qtts_talker_generate).
        pg = pg < pps ? pg : pps - 1;
        const int page = CT ? b * pps + pg : p.kv.page_table[b * pps + pg];
        return (((size_t)p.layer * p.kv.n_pages + page) * p.kv.nkv + kvh) * (16 * HD);
    };
    // K fragments of block `blk`: tile A rows = keys {0-3, 8-11} of both pages, tile B rows = keys {4-7, 12-15}; lane (row lj, lq)
    // holds dims 32 t + 8 lq .. + 8 for k-step t.  V fragments: dim block d (16 dims), lane (dim 16 d + lj, lq) holds keys
    // 8 lq .. 8 lq + 7 of the block = keys 8 (lq & 1) .. + 8 of page 2 blk + (lq >> 1).
    auto load_block = [&](u32x4 (&kA)[4], u32x4 (&kB)[4], u32x4 (&vT)[8], int blk) {
        const size_t pk = page_base(2 * blk + (lj >> 3));                  // rows 0-7: first page, rows 8-15: second page
        const int kin = ((lj >> 2) & 1) * 8 + (lj & 3);                    // key inside the page for tile A (tile B: + 4)
        const u32x4* ka = reinterpret_cast<const u32x4*>(kc + pk + (size_t)kin * HD + lq * 8);
        const u32x4* kb = reinterpret_cast<const u32x4*>(kc + pk + (size_t)(kin + 4) * HD + lq * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t) { kA[t] = ka[t * 4]; kB[t] = kb[t * 4]; }     // (32 dims = 64 B = 4 x 16-B units per k-step)
        const size_t pv = page_base(2 * blk + (lq >> 1));
        const u32x4* vb = reinterpret_cast<const u32x4*>(vc + pv + (size_t)lj * 16 + (lq & 1) * 8);
#pragma unroll
        for (int d = 0; d < 8; ++d) vT[d] = vb[d * 32];                    // (16 dims x 16 keys x 2 B = 512 B = 32 units per dim block)
    };

    // split-KV: workgroup y of gridDim.y handles the 32-key blocks [b0s, b0s + bps) of the key span p.max_len
    const int nsplit = gridDim.y, split = blockIdx.y;
    const int bps = (((p.max_len + 31) >> 5) + nsplit - 1) / nsplit;
    const int b0s = split * bps;
    // ---- 0. this step's row first (loads return in request order), then the K / V blocks of the register window, speculatively
    float x0v[GQ + 2], x1v[GQ + 2];
#pragma unroll
    for (int vi = 0; vi < GQ + 2; ++vi) {
        const int col = vi < GQ ? (kvh * GQ + vi) * HD : (vi == GQ ? (p.nh + kvh) * HD : (p.nh + p.nkv + kvh) * HD);
        const float* src = p.qkv + (size_t)b * p.ld + col;
        x0v[vi] = src[lane]; x1v[vi] = src[lane + 64];
    }
    const float wq0 = p.qw[lane], wq1 = p.qw[lane + 64], wk0 = p.kw[lane], wk1 = p.kw[lane + 64];
    const float invf = p.inv_freq[lane];
    u32x4 kA[NB][4], kB[NB][4], vT[NB][8];
#pragma unroll
    for (int n = 0; n < NB; ++n) load_block(kA[n], kB[n], vT[n], b0s + wave + 4 * n);
    const int S0 = p.len_dev ? *p.len_dev : p.len_static;   // KV length before this step = position of the new key
    const int npad = p.n_pad ? p.n_pad[b] : 0;
    const int done = p.done_flag ? *p.done_flag : 0;
    if (done) return;
    QTTS_TS_DRAINED(1);
    const int bend = min(b0s + bps, (S0 + 31) >> 5);         // blocks of CACHED keys of this split end here

    // ---- 1. q / k RMSNorm + RoPE of the new token, per wave; K / V append by wave 0 of split 0 (V dim-major inside its page)
    const float ang = (float)(S0 - npad) * invf;
    const float cs = cosf(ang), sn = sinf(ang);
    float sq[GQ > 0 ? GQ : 1];
    float kx0 = 0.f, kx1 = 0.f;
#pragma unroll
    for (int vi = GQ; vi >= 0; --vi) {                       // k first (vi == GQ), then the query heads: the scores of the new key need it
        float x0 = x0v[vi], x1 = x1v[vi];
        const float ss = wave_sum64_dpp(x0 * x0 + x1 * x1);
        const float rs = rsqrtf(ss / (float)HD + p.eps);
        x0 = (vi < GQ ? wq0 : wk0) * (x0 * rs);
        x1 = (vi < GQ ? wq1 : wk1) * (x1 * rs);
        const float o0 = x0 * cs - x1 * sn, o1 = x1 * cs + x0 * sn;
        x0 = o0; x1 = o1;
        if (vi == GQ) {
            const bf16_t h0 = f32_to_bf16(x0), h1 = f32_to_bf16(x1);
            if (wave == 0 && split == 0 && (S0 >> 4) < pps) {
                bf16_t* cdst = reinterpret_cast<bf16_t*>(p.kv.k);
                const size_t o = page_base(S0 >> 4) + (size_t)(S0 & 15) * HD;
                cdst[o + lane] = h0; cdst[o + lane + 64] = h1;
            }
            kx0 = bf16_to_f32(h0); kx1 = bf16_to_f32(h1);    // every wave uses the rounded key, what a later step reads back
        } else {
            xw[wave][vi][lane] = x0; xw[wave][vi][lane + 64] = x1;
            sq[vi] = wave_sum64_dpp(x0 * kx0 + x1 * kx1) * rsqrtf((float)HD);        // score of the new key (fp32 q . k)
        }
    }
    {
        const bf16_t h0 = f32_to_bf16(x0v[GQ + 1]), h1 = f32_to_bf16(x1v[GQ + 1]);
        if (wave == 0 && split == 0 && (S0 >> 4) < pps) {
            bf16_t* cdst = reinterpret_cast<bf16_t*>(p.kv.v);
            const size_t o = page_base(S0 >> 4) + (S0 & 15);
            cdst[o + (size_t)lane * 16] = h0; cdst[o + (size_t)(lane + 64) * 16] = h1;
        }
        xw[wave][GQ + 1][lane] = bf16_to_f32(h0); xw[wave][GQ + 1][lane + 64] = bf16_to_f32(h1);
    }
    __builtin_amdgcn_wave_barrier();             // wave-private LDS slice: program order within the wave is all that is needed
    // B operand of S = K q^T: lane (head lj, lq) <- q[lj][32 t + 8 lq .. + 8] as bf16; columns >= GQ are zero
    u32x4 qB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(&xw[wave][lj < GQ ? lj : 0][32 * t + 8 * lq]);
        const float4 c = *reinterpret_cast<const float4*>(&xw[wave][lj < GQ ? lj : 0][32 * t + 8 * lq + 4]);
        u32x4 v;
        v[0] = pack_bf16(a.x, a.y); v[1] = pack_bf16(a.z, a.w); v[2] = pack_bf16(c.x, c.y); v[3] = pack_bf16(c.z, c.w);
        qB[t] = lj < GQ ? v : (u32x4){0u, 0u, 0u, 0u};
    }
    QTTS_TS_DRAINED(2);

    // ---- 2. online softmax over this wave's blocks; everything in registers
    const float scale = rsqrtf((float)HD);
    float m = -INFINITY, l = 0.f;                // of head lj (meaningful for lj < GQ); l: this lane's keys only until the end
    f32x4 acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto process = [&](const u32x4 (&ka)[4], const u32x4 (&kb)[4], const u32x4 (&vt)[8], int blk) {
        f32x4 sA = {0.f, 0.f, 0.f, 0.f}, sB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bf16x8 a, bq, c;
            *reinterpret_cast<u32x4*>(&a) = ka[t]; *reinterpret_cast<u32x4*>(&c) = kb[t]; *reinterpret_cast<u32x4*>(&bq) = qB[t];
            sA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, sA, 0, 0, 0);
            sB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c, bq, sB, 0, 0, 0);
        }
        // this lane: keys key0 + e, e = 0..7 (tile A: e = 0..3, tile B: e = 4..7) of head lj
        const int key0 = 32 * blk + 8 * lq;
        float sc[8];
        float mc = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = key0 + e;
            const bool valid = key < S0 && key >= npad;      // left-pad slots were never written; slots >= S0 hold nothing yet
            const float v = (e < 4 ? sA[e] : sB[e - 4]) * scale;
            sc[e] = valid ? v : -INFINITY;
            mc = fmaxf(mc, sc[e]);
        }
        mc = fmaxf(mc, __shfl_xor(mc, 16));
        mc = fmaxf(mc, __shfl_xor(mc, 32));
        const float mn = fmaxf(m, mc);
        // (mn == -inf: no valid key so far in this column; every exponent below is then exp(-inf - (-inf)) = NaN -> forced to 0)
        const float f = m > -INFINITY ? att_exp<bf16_t>(m - mn) : 0.f;
        float pr[8], ps = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { pr[e] = sc[e] > -INFINITY ? att_exp<bf16_t>(sc[e] - mn) : 0.f; ps += pr[e]; }
        l = l * f + ps;
        m = mn;
        u32x4 pb;
        pb[0] = pack_bf16(pr[0], pr[1]); pb[1] = pack_bf16(pr[2], pr[3]); pb[2] = pack_bf16(pr[4], pr[5]); pb[3] = pack_bf16(pr[6], pr[7]);
        // V fragments of never-written / not-yet-written keys may hold anything (NaN x 0 = NaN): mask them to zero.  The 8 keys of a
        // V fragment are keys 32 blk + 8 lq + e as well (same lq), so one mask serves all 8 dim blocks.
        u32x4 vm;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const int k0 = key0 + 2 * e2;
            vm[e2] = ((k0 < S0 && k0 >= npad) ? 0x0000ffffu : 0u) | ((k0 + 1 < S0 && k0 + 1 >= npad) ? 0xffff0000u : 0u);
        }
        bf16x8 pB;
        *reinterpret_cast<u32x4*>(&pB) = pb;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            bf16x8 va;
            *reinterpret_cast<u32x4*>(&va) = vt[d] & vm;
            acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pB, acc[d] * f, 0, 0, 0);
        }
    };
#pragma unroll
    for (int n = 0; n < NB; ++n)
        if (b0s + wave + 4 * n < bend) process(kA[n], kB[n], vT[n], b0s + wave + 4 * n);
    for (int blk = b0s + wave + 4 * NB; blk < bend; blk += 4 * NB) {        // beyond the register window: NB blocks per latency round
#pragma unroll
        for (int n = 0; n < NB; ++n)
            if (blk + 4 * n < bend) load_block(kA[n], kB[n], vT[n], blk + 4 * n);
#pragma unroll
        for (int n = 0; n < NB; ++n)
            if (blk + 4 * n < bend) process(kA[n], kB[n], vT[n], blk + 4 * n);
    }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    // ---- 3. merge of the 4 waves and of the new key (fixed order).  acc[d][r] of lane (head lj, lq) = dim 16 d + 4 lq + r
    if (lj < GQ) {
#pragma unroll
        for (int d = 0; d < 8; ++d) *reinterpret_cast<f32x4*>(&red[wave][lj][16 * d + 4 * lq]) = acc[d];
        if (lq == 0) { gm[wave][lj] = m; gl[wave][lj] = l; }
    }
    if (wave == 0 && lane == 0) {
#pragma unroll
        for (int qi = 0; qi < GQ; ++qi) snew[qi] = sq[qi];
    }
    QTTS_TS_DRAINED(3);
    __syncthreads();
    QTTS_TS(4);
    if (tid < GQ * HD) {
        const int qi = tid / HD, dd = tid % HD;
        // the split whose key range holds position S0 also owns the new key
        const bool has_new = split == min(nsplit - 1, (S0 >> 5) / bps);
        float mm = has_new ? snew[qi] : -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, gm[w][qi]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = gm[w][qi] > -INFINITY ? att_exp<bf16_t>(gm[w][qi] - mm) : 0.f;
            num += red[w][qi][dd] * f;
            den += gl[w][qi] * f;
        }
        if (has_new) {
            const float f = att_exp<bf16_t>(snew[qi] - mm);
            num += xw[0][GQ + 1][dd] * f;
            den += f;
        }
        if (nsplit > 1) {                                    // partial result of this split: numerator | max | denominator
            float* pp = p.part + (((size_t)blockIdx.x * nsplit + split) * GQ + qi) * (HD + 2);
            pp[dd] = num;
            if (dd == 0) { pp[HD] = mm; pp[HD + 1] = den; }
            QTTS_TS_DRAINED(5);
            QTTS_TS_END(attn, 2, S0, nsplit);
            return;
        }
        const size_t o = (size_t)b * p.ldo + (kvh * GQ + qi) * HD + dd;
        const float r = num / den;
        if (p.out_bf16) reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(r);
        else p.out[o] = r;
    }
    QTTS_TS_DRAINED(5);
    QTTS_TS_END(attn, 2, S0, nsplit);
}

// merge of the split-KV partial results (fixed order): out = sum_s num_s e^(m_s - m) / sum_s den_s e^(m_s - m)
template <int GQ>
__global__ __launch_bounds__(256) void attn_merge_kernel(AttnDecodeParams p) {
    constexpr int HD = 128;
    if (p.done_flag && *p.done_flag) return;
    const int b = blockIdx.x / p.nkv, kvh = blockIdx.x % p.nkv;
    const int tid = threadIdx.x;
    if (tid >= GQ * HD) return;
    const int qi = tid / HD, dd = tid % HD;
    const float* base = p.part + ((size_t)blockIdx.x * p.nsplit * GQ + qi) * (HD + 2);
    const size_t stride = (size_t)GQ * (HD + 2);
    float mm = -INFINITY;
    for (int s = 0; s < p.nsplit; ++s) mm = fmaxf(mm, base[s * stride + HD]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < p.nsplit; ++s) {
        const float ms = base[s * stride + HD];
        const float f = ms > -INFINITY ? expf(ms - mm) : 0.f;
        num += base[s * stride + dd] * f;
        den += base[s * stride + HD + 1] * f;
    }
    const size_t o = (size_t)b * p.ldo + (kvh * GQ + qi) * HD + dd;
    const float r = num / den;
    if (p.out_bf16) reinterpret_cast<bf16_t*>(p.out)[o] = f32_to_bf16(r);
    else p.out[o] = r;
}

template <typename KVT, int GQ>
static void launch_attn_tk_c(const AttnDecodeParams& p, dim3 grid, hipStream_t st) {
    if (p.kv.contig) hipLaunchKernelGGL((attn_tk_kernel<KVT, GQ, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_tk_kernel<KVT, GQ, false>), grid, dim3(256), 0, st, p);
}
static void launch_attn_tk(const AttnDecodeParams& p, int GQ, dim3 grid, hipStream_t st) {
    if (p.kv.vt) {               // bf16 cache with transposed V pages: both products on the matrix pipe
        QTTS_REQUIRE(p.kv.bf16, QTTS_ERR_ARG, "attn_decode: transposed V pages are a bf16-cache layout");
        if (p.kv.contig) { if (GQ == 1) hipLaunchKernelGGL((attn_tk16_kernel<1, true>), grid, dim3(256), 0, st, p);
                           else hipLaunchKernelGGL((attn_tk16_kernel<2, true>), grid, dim3(256), 0, st, p); }
        else { if (GQ == 1) hipLaunchKernelGGL((attn_tk16_kernel<1, false>), grid, dim3(256), 0, st, p);
               else hipLaunchKernelGGL((attn_tk16_kernel<2, false>), grid, dim3(256), 0, st, p); }
        return;
    }
    if (p.kv.bf16) { if (GQ == 1) launch_attn_tk_c<bf16_t, 1>(p, grid, st); else launch_attn_tk_c<bf16_t, 2>(p, grid, st); }
    else { if (GQ == 1) launch_attn_tk_c<float, 1>(p, grid, st); else launch_attn_tk_c<float, 2>(p, grid, st); }
}
// cos | sin table for launch-time-known positions: out[pos][0][i] = cosf(pos * inv_freq[i]), out[pos][1][i] = sinf(...), i < 64 -- the
// same expression, compiled by the same compiler, as in the attention kernels
__global__ void rope_table_kernel(const float* inv_freq, int n_pos, float* out) {
    const int pos = blockIdx.x, i = threadIdx.x;
    if (pos >= n_pos || i >= 64) return;
    const float ang = (float)pos * inv_freq[i];
    out[(size_t)pos * 128 + i] = cosf(ang);
    out[(size_t)pos * 128 + 64 + i] = sinf(ang);
}
void launch_rope_table(const float* inv_freq, int n_pos, float* out, hipStream_t st) {
    hipLaunchKernelGGL(rope_table_kernel, dim3(n_pos), dim3(64), 0, st, inv_freq, n_pos, out);
    QTTS_CHECK_HIP(hipGetLastError());
}

template <typename KVT, int NQ, bool CT>
static void launch_attn_decode_c(const AttnDecodeParams& p, size_t lds, hipStream_t st);
template <typename KVT, int NQ>
static void launch_attn_decode_t(const AttnDecodeParams& p, size_t lds, hipStream_t st) {
    if (p.kv.contig) launch_attn_decode_c<KVT, NQ, true>(p, lds, st);
    else launch_attn_decode_c<KVT, NQ, false>(p, lds, st);
}
template <typename KVT, int NQ, bool CT>
static void launch_attn_decode_c(const AttnDecodeParams& p, size_t lds, hipStream_t st) {
    auto kern = attn_decode_kernel<KVT, NQ, CT>;
    if (lds > 48 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 150 * 1024);
    hipLaunchKernelGGL(kern, dim3(p.B * p.nkv), dim3(256), lds, st, p);
}

void launch_attn_decode(const AttnDecodeParams& p, hipStream_t st) {
    QTTS_REQUIRE(p.hd == 128, QTTS_ERR_ARG, "attn_decode: head_dim must be 128");
    const int GQ = p.nh / p.nkv, NQ = p.n_new * GQ;
    QTTS_REQUIRE((NQ == 1 || NQ == 2 || NQ == 4) && p.n_new <= 2, QTTS_ERR_ARG, "attn_decode: 1, 2 or 4 queries per kv head");
    QTTS_REQUIRE(!p.kv.vt || (p.n_new == 1 && GQ <= 2 && (p.len_dev || p.n_pad || p.len_static + 1 > 16)), QTTS_ERR_ARG,
                 "attn_decode: a transposed-V cache is read by the talker's single-token kernel only");
    if (p.n_new == 2 && !p.len_dev && !p.n_pad && p.len_static == 0 && GQ <= 2) {          // the code predictor's pass 0
        const dim3 grid(p.B * p.nkv);
        if (p.kv.bf16) { if (p.kv.contig) hipLaunchKernelGGL((attn_cp0_kernel<bf16_t, true>), grid, dim3(256), 0, st, p);
                         else hipLaunchKernelGGL((attn_cp0_kernel<bf16_t, false>), grid, dim3(256), 0, st, p); }
        else { if (p.kv.contig) hipLaunchKernelGGL((attn_cp0_kernel<float, true>), grid, dim3(256), 0, st, p);
               else hipLaunchKernelGGL((attn_cp0_kernel<float, false>), grid, dim3(256), 0, st, p); }
        QTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    if (p.n_new == 1 && !p.len_dev && !p.n_pad && p.len_static + 1 <= 16 && GQ <= 2) {     // the code predictor's passes >= 1
        const dim3 grid(p.B * p.nkv);
        if (p.kv.bf16) { if (p.kv.contig) hipLaunchKernelGGL((attn_cp_kernel<bf16_t, true>), grid, dim3(256), 0, st, p);
                         else hipLaunchKernelGGL((attn_cp_kernel<bf16_t, false>), grid, dim3(256), 0, st, p); }
        else { if (p.kv.contig) hipLaunchKernelGGL((attn_cp_kernel<float, true>), grid, dim3(256), 0, st, p);
               else hipLaunchKernelGGL((attn_cp_kernel<float, false>), grid, dim3(256), 0, st, p); }
        QTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    if (p.n_new == 1 && GQ <= 2) {            // the talker's single-token step: any length, any padding (attn_tk above)
        const int ns = p.nsplit > 1 ? p.nsplit : 1;
        QTTS_REQUIRE(ns == 1 || p.part, QTTS_ERR_ARG, "attn_decode: split-KV needs the partial-result buffer");
        const dim3 grid(p.B * p.nkv, ns);
        launch_attn_tk(p, GQ, grid, st);
        if (ns > 1) {
            if (GQ == 1) hipLaunchKernelGGL((attn_merge_kernel<1>), dim3(p.B * p.nkv), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((attn_merge_kernel<2>), dim3(p.B * p.nkv), dim3(256), 0, st, p);
        }
        QTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    const size_t lds = ((size_t)NQ * 128 + 2 * p.n_new * 128 + 16 * NQ * 128 + (size_t)NQ * p.max_len + 8) * sizeof(float);
    QTTS_REQUIRE(lds <= 150 * 1024, QTTS_ERR_LIMIT, "attn_decode: max_len too large for LDS scores");
    if (p.kv.bf16) {
        if (NQ == 1) launch_attn_decode_t<bf16_t, 1>(p, lds, st);
        else if (NQ == 2) launch_attn_decode_t<bf16_t, 2>(p, lds, st);
        else launch_attn_decode_t<bf16_t, 4>(p, lds, st);
    } else {
        if (NQ == 1) launch_attn_decode_t<float, 1>(p, lds, st);
        else if (NQ == 2) launch_attn_decode_t<float, 2>(p, lds, st);
        else launch_attn_decode_t<float, 4>(p, lds, st);
    }
    QTTS_CHECK_HIP(hipGetLastError());
}

// =================================================================================== cp_attn_o (round 4)
// The code predictor's passes >= 1: a layer's q|k|v GEMM (template QKV; layers >= 1 -- layer 0's row comes from the table), attention AND
// o-projection in one launch (70 launches per frame, 56 of them with the GEMM in front).  As separate launches they cost, on the in-kernel
// clock, q|k|v 2.7 + attn_cp 2.4 + the 4 MB o-projection 2.7 us and a 1.7-us boundary each = 13.0 us; here ~7.6 + 1.6 us
// (profiles/r04_cp_attn_o.md: eight builds, what each timeline said).  The attention output is the o-projection's k dimension, head by
// head, so the GEMM is split over k BY KV HEAD and every edge between workgroups is narrow:
//   workgroup (row pair, kv head g, 128-feature chunk c), 4 waves = 256 workgroups at batch 8: one wave per SIMD of the chip.
//   QKV front: the workgroup first computes 16-feature strip `blockIdx` of q|k|v = rsqrt(mean x^2 + eps) W' x (RMSNorm weight folded into W';
//     each wave a quarter of k: 8 MFMAs, the row variances from the same bf16 x fragments; quarters added in wave order) and stores it as
//     tagged granules.  Then wave (sequence, query head hh of the kv head) reads its q / k / v rows back -- parts of six other workgroups'
//     strips -- and
//   attention: runs that head's attention -- the arithmetic of attn_cp, statement for statement (q / k RMSNorm + RoPE at the static
//     position, K / V rounded through the cache type, 16 key slots, fp32 softmax, PV) -- out of a wave-private LDS slice; only chunk 0
//     appends K / V.  The eight chunk workgroups of a (row pair, kv head) repeat that attention (a few KB of reads each) instead of
//     exchanging it: the attention stage is ~400 VALU instructions per (sequence, head), so the repeats must not share a SIMD (second
//     build, 16 waves per workgroup on 64 CUs: 2.3 us of VALU queueing behind the barrier).
//   o-projection: the 2 x 256 bf16 result is the B operand (two columns of the MFMA tile) of 16 MFMAs per wave against this workgroup's
//     128 x 256 block of Wo (wave = two 16-feature strips; requested at kernel entry, or behind the strip's store with the front: they
//     stream while the workgroup waits and attends); the partial sums leave as tagged granules; the workgroup of the LAST kv head is the
//     reducer of its (row pair, chunk): its own partial sum stays in LDS, it reads the other seven slabs, adds the eight in kv-head order
//     + the residual and writes the hidden state (fp32 + bf16 copy).  Fixed summation orders: the result does not depend on timing.
//   hand-off WITHOUT a ticket or a fence: every value is an 8-byte granule {fp32 value, launch tag}, stored write-through (sc1), read with
//     sc1 loads until it carries the tag.  Tag = (frame serial << 7 | launch slot): nobody in the launch writes the word it derives from,
//     and no launch that wrote the buffers shortly before had the same pair.  A consumer waits ~0.4 us before its first read (a read
//     issued at once comes back stale and costs a round trip: third build) and keeps TWO reads in flight ~0.1 us apart (the producers
//     do not store at the same instant: with single reads the frame time moved 5 % with the delay).  (First build: partial sums +
//     drained stores + an arrival ticket + last-arriver reduction = 3.2 us of hand-off: what a boundary costs.)
//   A consumer cannot hang the device: after GRANULE_SPIN_LIMIT re-reads it gives up (cpao_give_up below).  All 256 workgroups
//   are resident from the start and producers never wait, so there is no circular wait inside a launch; across concurrent launches see
//   talker_engine.hip: fused_admit (engines are admitted per device by an account of the register file).
// bf16 cache, two query heads per kv head, head_dim 128, batch <= 8 only; everything else keeps attn_cp + the decode GEMM.

// A consumer that never saw its producers' tag: raise the engine's flag AND latch the generation's stop flag -- every later kernel of
// the frame chain (and every later frame step of the burst) returns at its `done` check, so one lost launch costs one give-up, the
// garbage this launch still writes is never consumed, and the host finds the flag at its next poll (talker_engine.hip: generate /
// stream_step / stream_end raise QTTS_ERR_STATE).  Only this cold block differs from round 4's kernel: a first version that also
// tracked "gave up" per lane (to poison what it hands on) changed the code of the polling loops -- the two reads in flight collapsed
// into one -- and cost 1.3 us per launch on the MI355X (profiles/r05_cp_attn_o_giveup_ab.md).
__device__ __forceinline__ void cpao_give_up(const CpAttnOParams& P) {
    if (P.err) __hip_atomic_store(P.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (P.done_latch) __hip_atomic_store(P.done_latch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The operands of the FIRST requests are leading scalar arguments: with -amdgpu-kernarg-preload-count they arrive in SGPRs with the wave
// instead of behind an s_load round trip of the by-value struct (as the decode GEMM's, profiles/r03_ab_kpre.md).
#define QTTS_CPAO_ARGS(P) ((P).Wqkv ? (P).Wqkv : static_cast<const void*>((P).a.qkv)), (P).x16, (P).serial, (P).a.done_flag, (P).a.B, (P).ldx16, (P).K, (P).slot, (P)
// F32 (round 5: the exact parity mode's instantiation, the construction's bit-exact leg): fp32 operators in the fp32 decode GEMM's packing
// (k-tiles of 16, four v_mfma_f32_16x16x4_f32 each), fp32 x rows (`x16` then points to floats), fp32 cache, the attention output stays
// fp32 in the B tile -- the arithmetic of the fp32 engines' three launches in the fused launch's summation orders.
template <bool CT, bool QKV, bool F32>
__global__ __launch_bounds__(256) void cp_attn_o_kernel(const void* k0, const unsigned short* kx16, const int* kserial, const int* kdone, int kB, int kldx16,
                                                        int kK, int kslot, CpAttnOParams P) {
    if constexpr (QKV) P.Wqkv = k0; else P.a.qkv = static_cast<const float*>(k0);
    P.x16 = kx16; P.serial = kserial; P.a.done_flag = kdone; P.a.B = kB; P.ldx16 = kldx16; P.K = kK; P.slot = kslot;
    constexpr int HD = 128, MAXK = 16, KW = F32 ? 8 : 4, NKV = 8, BSTR = 264;       // BSTR: elements per row of the B tile (16-B rows, bank-spread)
    constexpr int KT = F32 ? 16 : 32;                   // k per tile of the packed operators
    constexpr int NKS = F32 ? 16 : 8;                   // k-tiles per wave: a quarter of K = 1024 in the front, the 256 k of a kv head in the o-projection
    typedef typename CpaoKv<F32>::type KVT;
    // ONE LDS object (a second one de-pipelines the loads around it):
    //   [4 waves][q | kn | vn : 128 floats each] | B tile [2][BSTR] bf16 | the reducer's own partial sum [2][128] floats
    //   | QKV: the four k quarters of the q|k|v strip [4 waves][64 lanes][4] floats and of the row sums of squares [4][16]
    constexpr int WS_BYTES = 4 * 1536, BT_BYTES = 2 * BSTR * (F32 ? 4 : 2), OWN_BYTES = 2 * 128 * 4, QP_BYTES = QKV ? 4 * 64 * 16 + 4 * 16 * 4 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[WS_BYTES + BT_BYTES + OWN_BYTES + QP_BYTES];
    const AttnDecodeParams& p = P.a;
    const int nchunk = P.H >> 7;
    // blockIdx = (row pair, kv head, chunk), chunk fastest: with 8 chunks the 32 workgroups that read one chunk's columns of Wo share an XCD's L2
    const int rq = blockIdx.x / (NKV * nchunk), gc = blockIdx.x - rq * (NKV * nchunk);
    const int g = gc / nchunk, c = gc - g * nchunk;
    if (!QKV && rq * 2 >= p.B) return;                 // (no sequence in this row pair: nobody waits for this workgroup either)
    QTTS_TS_BEGIN();                       // (tstamp build: 1 = every request has arrived, 2 = attention done, 3 = partial sums stored, 4 = the other slabs read, 5 = reduced)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rr = wave & 1, hh = wave >> 1;         // attention: sequence 2 rq + rr, query head 2 g + hh; GEMM: strips 2 wave, 2 wave + 1 of the chunk
    const int kk = lane >> 2, qq = lane & 3;
    const int S0 = p.len_static, S1 = S0 + 1;
    const int row_w = rq * 2 + rr;
    const bool have = row_w < p.B;
    const int b = have ? row_w : 0;
    const KVT* kc = reinterpret_cast<const KVT*>(p.kv.k);
    const KVT* vc = reinterpret_cast<const KVT*>(p.kv.v);
    auto key_base = [&](int s) -> size_t {
        const int page = CT ? b * p.kv.pages_per_seq + (s >> 4) : p.kv.page_table[b * p.kv.pages_per_seq + (s >> 4)];
        return ((((size_t)p.layer * p.kv.n_pages + page) * p.kv.nkv + g) * 16 + (s & 15)) * HD;
    };
    // ---- 0. every request of the attention stage, then this wave's block of Wo (two 16-feature strips x the 256 k of kv head g: 16 KB)
    // the launch tag: nobody in this launch writes the word it derives from, and the launches that wrote the granule buffers before this
    // one had another (serial, slot) pair -- a granule that carries the tag is this launch's
    const unsigned tag = ((unsigned)*P.serial << 7) | (unsigned)P.slot;
    const int li = lane & 15, lq = lane >> 4;
    // ---- 0a (QKV). this workgroup's strip of the layer's q|k|v GEMM: 16 features x K, the four waves take a quarter of k each.  Requested
    // first: every other workgroup's attention waits for these sums.
    cu32x4 gw[NKS], gx[NKS];
    if (QKV && P.phase != 1) {
        const int nkt = P.K / KT, kq = nkt >> 2;           // k-tiles; per wave kq of them (NKS at K = 1024)
        const cu32x4* wsrc = reinterpret_cast<const cu32x4*>(P.Wqkv) + ((size_t)blockIdx.x * nkt + wave * kq) * 64 + lane;
        const cu32x4* xsrc = F32 ? reinterpret_cast<const cu32x4*>(reinterpret_cast<const float*>(P.x16) + (size_t)(li < p.B ? li : 0) * P.ldx16 + wave * kq * 16 + lq * 4)
                                 : reinterpret_cast<const cu32x4*>(P.x16 + (size_t)(li < p.B ? li : 0) * P.ldx16 + wave * kq * 32 + lq * 8);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { gw[ks] = wsrc[ks * 64]; gx[ks] = xsrc[ks * 4]; }
    }
    const float* xrow = p.qkv + (size_t)b * p.ld;
    float xq[2], xk[2], xv[2];
    if constexpr (!QKV) {
        xq[0] = xrow[(g * 2 + hh) * HD + lane]; xq[1] = xrow[(g * 2 + hh) * HD + lane + 64];
        xk[0] = xrow[(p.nh + g) * HD + lane]; xk[1] = xrow[(p.nh + g) * HD + lane + 64];
        xv[0] = xrow[(p.nh + p.nkv + g) * HD + lane]; xv[1] = xrow[(p.nh + p.nkv + g) * HD + lane + 64];
    }
    const float qw0 = p.qw[lane], qw1 = p.qw[lane + 64], kw0 = p.kw[lane], kw1 = p.kw[lane + 64], invf = p.inv_freq[lane];
    const bool rtab = p.rope_cs && S0 < p.rope_cs_n;
    const float* rrow = rtab ? p.rope_cs + (size_t)S0 * 128 : p.inv_freq;
    const float ctab = rrow[lane], stab = rrow[rtab ? 64 + lane : lane];
    struct alignas(2 * sizeof(KVT)) VPair { KVT a, b; };
    cu32x4 kr[KW];
    VPair vr[MAXK];
    {
        const cu32x4* ksrc = reinterpret_cast<const cu32x4*>(kc + key_base(kk < S0 ? kk : 0) + qq * 32);
#pragma unroll
        for (int w = 0; w < KW; ++w) kr[w] = ksrc[w];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) vr[k] = *reinterpret_cast<const VPair*>(vc + key_base(k < S0 ? k : 0) + 2 * lane);
    }
    cu32x4 wf[2][NKS];
    auto load_wo = [&] {
        const int nkt = (p.nh * HD) / KT;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const cu32x4* wsrc = reinterpret_cast<const cu32x4*>(P.Wo) + ((size_t)(c * 8 + wave * 2 + s) * nkt + g * NKS) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) wf[s][ks] = wsrc[ks * 64];
        }
    };
    if constexpr (!QKV) load_wo();                      // (with the q|k|v front: requested behind the strip's store -- 64 KB that would delay the strip)
    const int done = p.done_flag ? *p.done_flag : 0;
    if (done) return;
    if (QKV && P.phase != 1) {
        // ---- 0b. 8 MFMAs per wave (D[feature 4 q + j][sequence i]), the row sums of squares from the same x fragments (the RMSNorm weight is
        // folded into Wqkv, so out = rsqrt(mean x^2 + eps) * W' x as in the decode GEMM), the four k quarters added in wave order
        f32x4 qa = (f32x4){0.f, 0.f, 0.f, 0.f};
        float ssq = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            cu32x4 xv4 = gx[ks];
            if (li >= p.B) xv4 = (cu32x4){0u, 0u, 0u, 0u};
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
        ssq += __shfl_xor(ssq, 32);                      // every lane: its row's sum over this wave's k quarter
        f32x4* qpart = reinterpret_cast<f32x4*>(smem + WS_BYTES + BT_BYTES + OWN_BYTES);
        float* qss = reinterpret_cast<float*>(smem + WS_BYTES + BT_BYTES + OWN_BYTES + 4 * 64 * 16);
        qpart[wave * 64 + lane] = qa;
        if (lq == 0) qss[wave * 16 + li] = ssq;
        __syncthreads();
        if (wave == 0 && li < p.B) {
            const f32x4 s4 = ((qpart[lane] + qpart[64 + lane]) + qpart[128 + lane]) + qpart[192 + lane];
            const float ss = ((qss[li] + qss[16 + li]) + qss[32 + li]) + qss[48 + li];
            const float rs = rsqrtf(ss / (float)P.K + P.eps_in);
            const WtBuf qg = wt_buf(P.qkv_gran, (size_t)8 * p.ld * 8);
            const int off = (int)(((size_t)li * p.ld + blockIdx.x * 16 + lq * 4) * 8);
            wt_store16(qg, off, (cu32x4){__float_as_uint(s4[0] * rs), tag, __float_as_uint(s4[1] * rs), tag});
            wt_store16(qg, off + 16, (cu32x4){__float_as_uint(s4[2] * rs), tag, __float_as_uint(s4[3] * rs), tag});
        }
        if (P.phase == 0) return;
    }
    if constexpr (QKV) {
        if (rq * 2 >= p.B) return;                     // (no sequence in this row pair: nothing to attend to, nobody waits for more)
        load_wo();                                     // arrives while this workgroup waits for its rows and attends
        QTTS_TS(1);
        // ---- 0c. this wave's rows of q | k | v, from the strips of 6 other workgroups: wait ~0.5 us before the first read (as the reducer
        // below: a read issued right away comes back stale), then until every granule carries this launch's tag
        const WtBuf qg = wt_buf(P.qkv_gran, (size_t)8 * p.ld * 8);
        int cols[3] = {(g * 2 + hh) * HD, (p.nh + g) * HD, (p.nh + p.nkv + g) * HD};
        // TWO reads in flight, `poll_step` apart: a read that left before the granules were readable costs the difference to the next
        // one, not a round trip (single reads: 2.73 / 2.65 / 2.58 ms per frame with the first read 0.3 / 0.5 / 0.75 us after the store)
        uint2 gq[3][2], gn[3][2];
        auto load_rows = [&](uint2 (&d)[3][2]) {
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) d[v][h2] = wt_load8(qg, (int)(((size_t)b * p.ld + cols[v] + lane + 64 * h2) * 8));
        };
        wt_first_pause(P.first_pause);
        load_rows(gq);
        wt_first_pause(P.poll_step);
        load_rows(gn);
        for (int spins = 0;; ++spins) {
            bool fresh = true;
#pragma unroll
            for (int v = 0; v < 3; ++v) fresh = fresh && gq[v][0].y == tag && gq[v][1].y == tag;
            if (fresh) break;
            if (spins > GRANULE_SPIN_LIMIT) { cpao_give_up(P); break; }
#pragma unroll
            for (int v = 0; v < 3; ++v) { gq[v][0] = gn[v][0]; gq[v][1] = gn[v][1]; }
            wt_first_pause(P.poll_step);
            load_rows(gn);
        }

        xq[0] = __uint_as_float(gq[0][0].x); xq[1] = __uint_as_float(gq[0][1].x);
        xk[0] = __uint_as_float(gq[1][0].x); xk[1] = __uint_as_float(gq[1][1].x);
        xv[0] = __uint_as_float(gq[2][0].x); xv[1] = __uint_as_float(gq[2][1].x);
    } else {
        QTTS_TS_DRAINED(1);
    }

    float* ws = reinterpret_cast<float*>(smem + wave * 1536);          // q | kn | vn
    bf16_t* Bt = reinterpret_cast<bf16_t*>(smem + WS_BYTES);
    float* Btf = reinterpret_cast<float*>(smem + WS_BYTES);             // (F32: the B tile holds floats)
    float* own = reinterpret_cast<float*>(smem + WS_BYTES + BT_BYTES);
    if (have) {
        // ---- 1. q / k RMSNorm + RoPE at position S0, K / V through the cache type (attn_cp's stage 1)
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
            const KVT k0 = kv_cast<KVT>(xk[0]), k1 = kv_cast<KVT>(xk[1]), v0 = kv_cast<KVT>(xv[0]), v1 = kv_cast<KVT>(xv[1]);
            if (c == 0) {                                  // one workgroup per (row pair, kv head) appends: head 0's wave the K row, head 1's the V row
                if (hh == 0) { KVT* kd = reinterpret_cast<KVT*>(p.kv.k); kd[o + lane] = k0; kd[o + lane + 64] = k1; }
                else { KVT* vd = reinterpret_cast<KVT*>(p.kv.v); vd[o + lane] = v0; vd[o + lane + 64] = v1; }
            }
            ws[HD + lane] = kv_load(&k0); ws[HD + lane + 64] = kv_load(&k1);
            ws[2 * HD + lane] = kv_load(&v0); ws[2 * HD + lane + 64] = kv_load(&v1);
        }
        __builtin_amdgcn_wave_barrier();             // wave-private LDS slice: program order within the wave is all that is needed
        // ---- 2. this wave's query head over the 16 key slots
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
    QTTS_TS(2);
    __syncthreads();
    // ---- 3. partial o-projection: D[feature 4 q + j][sequence li] of two strips over the 256 k of this kv head (columns 0 / 1 of the MFMA tile)
    const int row = rq * 2 + li;                        // (meaningful for li < 2)
    const bool col_ok = li < 2 && row < p.B;
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
    const bool reducer = g == NKV - 1;
    const WtBuf slab = wt_buf(P.part, (size_t)NKV * 8 * P.H * 8);
    if (reducer) {
        if (li < 2) {
#pragma unroll
            for (int s = 0; s < 2; ++s) *reinterpret_cast<f32x4*>(own + li * 128 + (wave * 2 + s) * 16 + lq * 4) = acc[s];
        }
    } else if (col_ok) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = (int)((((size_t)g * 8 + row) * P.H + c * 128 + (wave * 2 + s) * 16 + lq * 4) * 8);
            wt_store16(slab, off, (cu32x4){__float_as_uint(acc[s][0]), tag, __float_as_uint(acc[s][1]), tag});
            wt_store16(slab, off + 16, (cu32x4){__float_as_uint(acc[s][2]), tag, __float_as_uint(acc[s][3]), tag});
        }
    }
    QTTS_TS(3);
#if QTTS_TSTAMP
#define QTTS_TS_CPAO(tail_)                                                                                              \
    if (tid == 0 && ((tail_) || blockIdx.x % 9 == 4)) {                                                                  \
        const unsigned i_ = atomicAdd(&qtts::ts_cnt_attn, 1u);                                                           \
        if (i_ < qtts::TS_CAP) {                                                                                         \
            qtts::TsRec r_;                                                                                              \
            for (int k_ = 0; k_ < 6; ++k_) r_.t[k_] = ts_[k_];                                                           \
            r_.kind = 4; r_.a = S0; r_.b = 0; r_.blk = (int)blockIdx.x | ((tail_) << 16);                                \
            qtts::ts_log_attn[i_] = r_;                                                                                  \
        }                                                                                                                \
    }
#else
#define QTTS_TS_CPAO(tail_)
#endif
    if (!reducer) { QTTS_TS_CPAO(0) return; }
    // ---- 4. the reducer of (row pair, chunk c): hidden = (sum over kv heads, in order) + residual
    __syncthreads();                                     // (its own partial sum is in LDS)
    if (tid < 128) {
        const int rw = rq * 2 + (tid >> 6), c2 = (tid & 63) * 2, col = c * 128 + c2;
        if (rw < p.B) {
            const float2 res = *reinterpret_cast<const float2*>(P.res + (size_t)rw * P.H + col);
            // the other workgroups stored when this one did, and a write-through store takes ~0.6 us to become readable: a read issued now
            // would come back stale and cost a second round trip (third version's timeline: reduced 2.7 us after its own partial sum)
            cu32x4 pa[NKV - 1], pn[NKV - 1];
            auto load_slabs = [&](cu32x4 (&d)[NKV - 1]) {
#pragma unroll
                for (int g2 = 0; g2 < NKV - 1; ++g2) d[g2] = wt_load16(slab, (int)((((size_t)g2 * 8 + rw) * P.H + col) * 8));
            };
            wt_first_pause(P.first_pause);
            load_slabs(pa);
            wt_first_pause(P.poll_step);
            load_slabs(pn);
            for (int spins = 0;; ++spins) {                 // (two reads in flight, as for the q | k | v rows)
                bool fresh = true;
#pragma unroll
                for (int g2 = 0; g2 < NKV - 1; ++g2) fresh = fresh && pa[g2][1] == tag && pa[g2][3] == tag;
                if (fresh) break;
                if (spins > GRANULE_SPIN_LIMIT) { cpao_give_up(P); break; }
#pragma unroll
                for (int g2 = 0; g2 < NKV - 1; ++g2) pa[g2] = pn[g2];
                wt_first_pause(P.poll_step);
                load_slabs(pn);
            }

            float s0 = __uint_as_float(pa[0][0]), s1 = __uint_as_float(pa[0][2]);
#pragma unroll
            for (int g2 = 1; g2 < NKV - 1; ++g2) { s0 += __uint_as_float(pa[g2][0]); s1 += __uint_as_float(pa[g2][2]); }
            s0 += own[(tid >> 6) * 128 + c2]; s1 += own[(tid >> 6) * 128 + c2 + 1];
            s0 += res.x; s1 += res.y;
            float2 o2; o2.x = s0; o2.y = s1;
            *reinterpret_cast<float2*>(P.out + (size_t)rw * P.H + col) = o2;
            if (P.out16) *reinterpret_cast<unsigned*>(P.out16 + (size_t)rw * P.H + col) = pack_bf16(s0, s1);
        }
    }
    QTTS_TS(4);
    QTTS_TS_DRAINED(5);
    QTTS_TS_CPAO(1)
}

bool cp_attn_o_takes(const AttnDecodeParams& a, int H) {
    return a.hd == 128 && a.nkv == 8 && a.nh == 16 && a.n_new == 1 && !a.len_dev && !a.n_pad && a.len_static >= 1 && a.len_static + 1 <= 16 &&
           a.B >= 1 && a.B <= 8 && !a.kv.vt && H % 128 == 0 && H >= 128;          // (bf16 cache: the bf16 instantiations; fp32 cache: the F32 ones)
}

// bench.py's roofline leg (qtts_talker_set_profile(1)): the next launch goes out through hipExtLaunchKernelGGL with the caller's event
// pair -- the kernel's own begin / end timestamps, as for the decode GEMM (skinny.hip: skinny_set_launch_events)
static thread_local hipEvent_t tl_cpao_ev_start = nullptr, tl_cpao_ev_stop = nullptr;
void cp_attn_o_set_launch_events(hipEvent_t start, hipEvent_t stop) { tl_cpao_ev_start = start; tl_cpao_ev_stop = stop; }
#define QTTS_CPAO_LAUNCH(kern, grid, st, ...)                                                                                  \
    do {                                                                                                                       \
        if (tl_cpao_ev_start) hipExtLaunchKernelGGL(kern, grid, dim3(256), 0, st, tl_cpao_ev_start, tl_cpao_ev_stop, 0, __VA_ARGS__);  \
        else hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, __VA_ARGS__);                                                    \
    } while (0)

int cp_attn_o_grid(int H) { return 4 * 8 * (H / 128); }

int cp_attn_o_blocks_per_cu(bool f32) {
#ifdef QTTS_HOST_EMU
    if (const char* e = QTTS_ENV("QTTS_HOSTEMU_CPAO_BLOCKS_PER_CU")) return atoi(e);      // (tests of the admission rule)
    return f32 ? 1 : 2;                                 // what the gfx950 build reports (the q|k|v-front instantiations: 176-180 registers, two waves per SIMD)
#else
    int best = 1 << 30;
    auto probe = [&](auto kern) {
        int n = 0;
        QTTS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, 0));
        best = std::min(best, n);
    };
    if (f32) {
        probe(cp_attn_o_kernel<true, true, true>); probe(cp_attn_o_kernel<false, true, true>);
        probe(cp_attn_o_kernel<true, false, true>); probe(cp_attn_o_kernel<false, false, true>);
    } else {
        probe(cp_attn_o_kernel<true, true, false>); probe(cp_attn_o_kernel<false, true, false>);
        probe(cp_attn_o_kernel<true, false, false>); probe(cp_attn_o_kernel<false, false, false>);
    }
    return best;
#endif
}

void launch_cp_attn_o(const CpAttnOParams& P, hipStream_t st) {
    QTTS_REQUIRE(cp_attn_o_takes(P.a, P.H), QTTS_ERR_ARG, "cp_attn_o: shape (16 / 8 heads of 128, one new token, <= 16 keys, batch <= 8)");
    const bool f32 = !P.a.kv.bf16;                       // (the cache type says which engine this is: fp32 operators, rows and cache go together)
    QTTS_REQUIRE(!f32 || !P.out16, QTTS_ERR_ARG, "cp_attn_o: no bf16 copy of the hidden rows in fp32 mode");
    QTTS_REQUIRE(P.Wo && P.res && P.out && P.part && P.serial && P.a.qw && P.a.kw && P.a.inv_freq, QTTS_ERR_ARG, "cp_attn_o: null operand");
    QTTS_REQUIRE(P.slot >= 0 && P.slot < 128, QTTS_ERR_ARG, "cp_attn_o: slot must be 0..127");
    const dim3 grid(cp_attn_o_grid(P.H));                // (row pair, kv head, chunk)
    if (P.Wqkv) {      // with the layer's q|k|v GEMM in front: workgroup = one 16-feature strip of it, so the two grids must coincide
        QTTS_REQUIRE((int)grid.x * 16 == P.a.ld && P.K == 1024 && P.x16 && P.qkv_gran && P.ldx16 % 8 == 0, QTTS_ERR_ARG,
                     "cp_attn_o: the q|k|v front needs K = 1024, (nh + 2 nkv) * 128 == 16 * workgroups, x rows in the engine's type and the granule buffer");
        CpAttnOParams Q = P;
#ifdef QTTS_HOST_EMU
        // The emulator runs the workgroups of a launch one after the other; here every workgroup first produces and then waits for
        // other workgroups' strips, so the emulated launch runs as its two halves (the same code, both launches read the same tag).
        for (int ph = P.phase == 1 ? 1 : 0; ph < 2; ++ph) {     // (phase == 1 given: the consuming half alone -- the stale-granule test)
            Q.phase = ph;
#else
        {
            Q.phase = 2;
#endif
            if (f32) {
                if (P.a.kv.contig) QTTS_CPAO_LAUNCH((cp_attn_o_kernel<true, true, true>), grid, st, QTTS_CPAO_ARGS(Q));
                else QTTS_CPAO_LAUNCH((cp_attn_o_kernel<false, true, true>), grid, st, QTTS_CPAO_ARGS(Q));
            } else {
                if (P.a.kv.contig) QTTS_CPAO_LAUNCH((cp_attn_o_kernel<true, true, false>), grid, st, QTTS_CPAO_ARGS(Q));
                else QTTS_CPAO_LAUNCH((cp_attn_o_kernel<false, true, false>), grid, st, QTTS_CPAO_ARGS(Q));
            }
        }
    } else {
        QTTS_REQUIRE(P.a.qkv, QTTS_ERR_ARG, "cp_attn_o: null q|k|v rows");
        if (f32) {
            if (P.a.kv.contig) QTTS_CPAO_LAUNCH((cp_attn_o_kernel<true, false, true>), grid, st, QTTS_CPAO_ARGS(P));
            else QTTS_CPAO_LAUNCH((cp_attn_o_kernel<false, false, true>), grid, st, QTTS_CPAO_ARGS(P));
        } else {
            if (P.a.kv.contig) QTTS_CPAO_LAUNCH((cp_attn_o_kernel<true, false, false>), grid, st, QTTS_CPAO_ARGS(P));
            else QTTS_CPAO_LAUNCH((cp_attn_o_kernel<false, false, false>), grid, st, QTTS_CPAO_ARGS(P));
        }
    }
    QTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace qtts
