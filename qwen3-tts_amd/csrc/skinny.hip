// skinny.hip -- weight-streaming "skinny" GEMM for the autoregressive decode step (gfx950).
//
//   out[M][N] = epi( rstd[m] * sum_k x[m][k] * W'[n][k] )      M = batch rows (<= 64),  W' = W . diag(g)
//
// Decode is HBM-bound (8 flop/B at batch 8): the kernel's job is to pull every weight byte across HBM exactly
// once, with as little fixed latency around that stream as possible -- a frame step is a chain of ~400 of these launches, so
// every serialized memory round trip and every workgroup barrier inside the kernel is paid 400x.
//   * weights are re-packed at bind time into 1-KiB tiles that ARE the MFMA A-operand image
//     ([N/FS strips][K/KT k-tiles][4 k-slices][FS features][16 B]); a wave's `global_load_dwordx4` reads up to 1 KiB fully
//     contiguous, non-temporal (measured: plain loads are 7.6 % slower per frame, profiles/r02_ab_variants.md), straight to
//     VGPRs (each byte is used once: no LDS round trip).  The RMSNorm weight g is folded into W at bind (W' = W.diag(g));
//     rstd[m] factors out of the dot product.
//   * MFMA roles are swapped w.r.t. the textbook: A = 16 output features x k, B = k x 16 batch rows (batch padded to 16), so D
//     holds 4 consecutive features per lane -> float4 epilogue stores.
//     bf16: v_mfma_f32_16x16x32_bf16; f32: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, parity mode).
//   * a workgroup = 8 (or 4) waves; k-tiles are dealt round-robin to the waves; every wave keeps two chunks of weight loads in
//     flight; partial sums are combined through LDS in a fixed order (deterministic, no atomics anywhere).
//   * bf16 mode (`skinny2_kernel`, round 2): NOTHING is staged through LDS before the MFMA loop.  Every wave fetches the x
//     fragments of ITS OWN k-tiles (the B operand: lane (row, k-slice) <- 16 B of the producer's bf16 copy of x) together with
//     the weight tiles of the same chunk, so x latency and weight latency overlap and no wave waits for another one before its
//     first MFMA.  The RMSNorm row variance rides on the matrix pipe: `acc_ss += mfma(x, x)` gives X.X^T per wave, whose
//     diagonal is sum_k x[m][k]^2 -- one extra MFMA per k-tile instead of a reduction pass and two workgroup barriers
//     (round 1: x staged via LDS-DMA, barrier, per-row reduction, barrier; 4-5 us of a 5.5 us launch were such fixed costs).
//     The only barrier left is the one in front of the cross-wave combine.
//   * everything the epilogue needs (residual, bias) is loaded at kernel entry, under the weight stream.
#include "common.h"
#include "kernels.h"
#include "tstamp.h"
#include "granule.h"
#include <hip/hip_ext.h>

QTTS_TS_UNIT(skinny)

// Perf-ablation hooks (tools/ablate_skinny.py) exist only in the `ablate` build variant (build.py VARIANTS, -DQTTS_ABLATE=1); in the
// product build the expressions are the constant 0 and the branches are gone.
#ifndef QTTS_ABLATE
#define QTTS_ABLATE 0
#endif
#if QTTS_ABLATE
#define QTTS_ABL(p, bit) ((p).ablate & (bit))
#else
#define QTTS_ABL(p, bit) 0
#endif

namespace qtts {

// Roofline leg of bench.py (qtts_talker_set_profile): when an event pair is set, the next decode-GEMM launch goes out through
// hipExtLaunchKernelGGL, which stamps the events with THIS kernel's own begin / end timestamps (the dispatch's completion-signal
// times -- what rocprofv3's kernel trace reports), so that every launch of the REAL frame step is timed on its own.
static thread_local hipEvent_t tl_ev_start = nullptr, tl_ev_stop = nullptr;
void skinny_set_launch_events(hipEvent_t start, hipEvent_t stop) { tl_ev_start = start; tl_ev_stop = stop; }
#define QTTS_SK_LAUNCH(kern, grid, block, lds, st, ...)                                                               \
    do {                                                                                                              \
        if (tl_ev_start) hipExtLaunchKernelGGL(kern, grid, block, lds, st, tl_ev_start, tl_ev_stop, 0, __VA_ARGS__);  \
        else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                             \
    } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <class T>
__device__ inline T skinny_wload(const T* ptr) { return __builtin_nontemporal_load(ptr); }

__device__ inline void skinny_store4(const SkinnyParams& p, int row, int col, const f32x4& o, bool shadow) {
    if (shadow && p.out16) {
        uint2 h; h.x = pack_bf16(o[0], o[1]); h.y = pack_bf16(o[2], o[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out16) + (size_t)row * p.ldo + col) = h;
    }
    if (p.out_bf16) {
        uint2 h; h.x = pack_bf16(o[0], o[1]); h.y = pack_bf16(o[2], o[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)row * p.ldo + col) = h;
    } else *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + col) = o;
}

// ------------------------------------------------------------------------------------------ bf16 (the benchmarked mode)
// FS = output features per strip (16 | 8 | 4).  Narrow strips put GEMMs with few output features on all 256 CUs (a CU pulls
// only ~25 GB/s); lanes with (lane & 15) >= FS carry no weights and their MFMA rows are ignored.
// MT = 16-row tiles of x (M <= 16 * MT), SPW = strips per workgroup (2 = a SwiGLU gate/up pair, or two plain strips).
// U = k-tiles per chunk; two chunks of loads are in flight per wave.  EXACT: every wave owns the same number of k-tiles and
// that number is a multiple of U (all shapes of the real models) -- the chunk loads are then unconditional, back-to-back
// requests; the generic variant guards the tail tile by tile (tiny test dimensions, odd K).
// NCH > 0 (EXACT only): the number of chunks is a compile-time constant (1..3) and the whole kernel is straight-line code -- no
// branch, hence no register merge (a merge copy of a load result waits for that load) between the request bursts.
template <int MT, int SPW, int NW, int FS, bool XB16, int U, bool EXACT, int NCH>
__global__ __launch_bounds__(NW * 64) void skinny2_kernel(SkinnyParams p) {
    constexpr int KT = 32;                                       // k per tile
    constexpr int NS = SPW * MT + MT;                            // accumulators per wave: the GEMM's + one X.X^T per m-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sk[];
    f32x4* red = reinterpret_cast<f32x4*>(smem_sk);              // [NW][NS][64]
    QTTS_TS_BEGIN();

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tile validity becomes a scalar branch
    const int lj = lane & 15, lq = lane >> 4;
    const int nkt = p.K / KT;
    const int my_tiles = (nkt - wave + NW - 1) / NW;             // tile = wave + NW*i  (0: this wave has nothing to do)
    const int nchunks = (my_tiles + U - 1) / U;
    const int strip0 = blockIdx.x * SPW;

    f32x4 acc[SPW][MT], acc_ss[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc_ss[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < SPW; ++s) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // The texture-address unit of a CU takes 64 B per clock: a 16-B-per-lane load costs 16 clocks when all 64 lanes are
    // active, and the ~32 requests per wave x 8 waves of a launch are a measurable part of it (round-2 ablation: +0.7 us per
    // 1024 of K with every lane loading).  So only the lanes that own a weight row (lj < FS: 16 / 32 / 64 lanes) and the lanes
    // that own an x row (row < M: 32 lanes at batch 8) issue requests -- under ONE exec mask per chunk and operand, not one
    // branch per load; the other lanes hold zeros, and what they feed to the MFMA only reaches output rows / columns nobody
    // stores.
    const bool wlane = lj < FS;
    const u32x4* wbase[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
        wbase[s] = reinterpret_cast<const u32x4*>(p.Wp) + ((size_t)(strip0 + s) * nkt) * (FS * 4) + lq * FS + (wlane ? lj : 0);
    // B operand of tile kt, m-tile m: lane (lj, lq) <- x[m*16 + lj][kt*32 + lq*8 .. +8]
    const unsigned short* xp16[MT];
    const float* xp32[MT];
    bool xlane[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        xlane[m] = m * 16 + lj < p.M;
        const int row = xlane[m] ? m * 16 + lj : 0;
        xp16[m] = reinterpret_cast<const unsigned short*>(p.x) + (size_t)row * p.ldx + lq * 8;
        xp32[m] = p.x + (size_t)row * p.ldx + lq * 8;
    }
    // perf ablation (DEBUG): "no weight stream" / "no x fetch" collapse the tile stride to 0 -- every request of a wave then hits
    // one resident line -- so that the instruction stream is unchanged
    const size_t wstep = QTTS_ABL(p, 8) ? 0 : (size_t)(FS * 4);
    const int xstep = QTTS_ABL(p, 2) ? 0 : KT;

    // (non-temporal: measured twice -- round 1 whole-frame +7.6 % with plain loads; round 2 per GEMM class, plain loads for the code
    // predictor's re-read 157 MB: no gain with, +3.5 % without a warm-up -- profiles/r02_ab_inproc_prefetch_temporal.json)
    auto wload = [&](const u32x4* ptr) -> u32x4 { return skinny_wload(ptr); };
    auto load_x = [&](int m, int kt) -> u32x4 {
        if constexpr (XB16) return *reinterpret_cast<const u32x4*>(xp16[m] + kt * xstep);
        else {    // fp32 x (no bf16 copy from the producer): converted here; not on the frame step's hot path
            const float4 a = *reinterpret_cast<const float4*>(xp32[m] + kt * xstep);
            const float4 b = *reinterpret_cast<const float4*>(xp32[m] + kt * xstep + 4);
            u32x4 t;
            t[0] = pack_bf16(a.x, a.y); t[1] = pack_bf16(a.z, a.w);
            t[2] = pack_bf16(b.x, b.y); t[3] = pack_bf16(b.z, b.w);
            return t;
        }
    };
    // (the operand registers are zeroed ONCE, below: lanes outside the masks never write them again, so a masked request is an
    // in-place update and needs no merge copy -- a copy would wait for the load it copies)
    auto load_chunk = [&](u32x4 (&w)[SPW][U], u32x4 (&xq)[MT][U], int c) {
        if constexpr (EXACT) {
            if (FS == 16 || wlane) {                             // one exec mask for all weight requests of the chunk
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int s = 0; s < SPW; ++s) w[s][u] = wload(wbase[s] + (size_t)(wave + NW * (c * U + u)) * wstep);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (xlane[m]) {                                  // one exec mask per m-tile for its x requests
#pragma unroll
                    for (int u = 0; u < U; ++u) xq[m][u] = load_x(m, wave + NW * (c * U + u));
                }
        } else {
            const int n = my_tiles - c * U;                      // wave-uniform: tiles of this chunk that exist
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < n) {
                    const int kt = wave + NW * (c * U + u);
                    if (wlane) {
#pragma unroll
                        for (int s = 0; s < SPW; ++s) w[s][u] = wload(wbase[s] + (size_t)kt * wstep);
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        if (xlane[m]) xq[m][u] = load_x(m, kt);
                }
            }
        }
    };

    // ---- 1. the weight stream and this wave's x fragments start first.  Straight-line kernels (NCH = 1..3 chunks): EVERY chunk
    // has its own registers and is requested right here (up to 24 KiB per wave in flight); the generic kernel keeps two chunks
    // in flight and ping-pongs.
    constexpr int NSET = NCH > 0 ? NCH : 2;
    u32x4 wR[NSET][SPW][U], xR[NSET][MT][U];
#pragma unroll
    for (int k = 0; k < NSET; ++k)
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int s = 0; s < SPW; ++s) wR[k][s][u] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int m = 0; m < MT; ++m) xR[k][m][u] = (u32x4){0u, 0u, 0u, 0u};
        }
    if constexpr (NCH > 0) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) load_chunk(wR[k], xR[k], k);
    } else {
        load_chunk(wR[0], xR[0], 0);
        if (!EXACT || nchunks > 1) load_chunk(wR[1], xR[1], 1);
    }

    // ---- 2. epilogue operands of wave 0 are fetched now, under the weight stream
    f32x4 resv[SPW][MT], biasv[SPW];
    const bool epi_loads = wave == 0 && !QTTS_ABL(p, 4);
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int col = (p.act == ACT_SWIGLU ? blockIdx.x * 16 : (strip0 + s) * FS) + lq * 4;
        biasv[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (epi_loads && p.bias && lq * 4 < FS) biasv[s] = *reinterpret_cast<const f32x4*>(p.bias + (strip0 + s) * FS + lq * 4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            resv[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = m * 16 + lj;
            if (epi_loads && p.res && row < p.M && lq * 4 < FS && (p.act != ACT_SWIGLU || s == 0))
                resv[s][m] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
        }
    }
    const int done = (p.done_flag && !QTTS_ABL(p, 1)) ? *p.done_flag : 0;
    QTTS_TS(1);                            // every request of the straight-line kernels has been issued (and `done` has arrived)
    if (done) return;
    QTTS_TS_DRAINED(2);                    // ... and has arrived

    auto compute_tile = [&](u32x4 (&w)[SPW][U], u32x4 (&xq)[MT][U], int u) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            bf16x8 xb;
            *reinterpret_cast<u32x4*>(&xb) = xq[m][u];
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                bf16x8 wa;
                *reinterpret_cast<u32x4*>(&wa) = w[s][u];
                acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[s][m], 0, 0, 0);
            }
            if (p.norm) acc_ss[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, xb, acc_ss[m], 0, 0, 0);
        }
    };
    auto compute_chunk = [&](u32x4 (&w)[SPW][U], u32x4 (&xq)[MT][U], int c) {
        if constexpr (EXACT) {
#pragma unroll
            for (int u = 0; u < U; ++u) compute_tile(w, xq, u);
        } else {
            const int n = my_tiles - c * U;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u < n) compute_tile(w, xq, u);
        }
    };

    // ---- 3. consume
    if constexpr (NCH > 0) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) compute_chunk(wR[k], xR[k], k);
    } else {                               // ping-pong: the chunk after next is requested as soon as a buffer frees up
        for (int c = 0; c < nchunks; c += 2) {
            compute_chunk(wR[0], xR[0], c);
            if (c + 2 < nchunks) load_chunk(wR[0], xR[0], c + 2);
            if (c + 1 < nchunks) compute_chunk(wR[1], xR[1], c + 1);
            if (c + 3 < nchunks) load_chunk(wR[1], xR[1], c + 3);
        }
    }

    // ---- 4. cross-wave combine (fixed order) and epilogue by wave 0: the kernel's only barrier
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) red[(wave * NS + s * MT + m) * 64 + lane] = acc[s][m];
        red[(wave * NS + SPW * MT + m) * 64 + lane] = acc_ss[m];
    }
    QTTS_TS_DRAINED(3);                    // MFMAs done, partial sums in LDS
    __syncthreads();
    QTTS_TS(4);
    if (wave != 0) return;

    // v[s][r] = out[row = m*16 + lj][feature = (strip0+s)*FS + lq*4 + r]
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + lj;
        float rstd = 1.f;
        if (p.norm) {
            // X.X^T: lane (j, q) component r holds entry (i = 4q + r, j); the diagonal element of row lj sits in lane
            // (lj, lj >> 2), component lj & 3
            const float* rf = reinterpret_cast<const float*>(red);
            const int src = (((lj >> 2) * 16 + lj) << 2) + (lj & 3);
            float ssum = rf[((0 * NS + SPW * MT + m) * 64) * 4 + src];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) ssum += rf[((w2 * NS + SPW * MT + m) * 64) * 4 + src];     // fixed order
            rstd = rsqrtf(ssum / (float)p.K + p.eps);
        }
        f32x4 v[SPW];
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            f32x4 t = red[(0 * NS + s * MT + m) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) t += red[(w2 * NS + s * MT + m) * 64 + lane];
            v[s] = t * rstd + biasv[s];
        }
        if (row >= p.M || lq * 4 >= FS) continue;
        if (p.act == ACT_SWIGLU) {
            if constexpr (SPW == 2) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (v[0][r] / (1.f + expf(-v[0][r]))) * v[1][r];
                o += resv[0][m];
                skinny_store4(p, row, blockIdx.x * 16 + lq * 4, o, false);
            }
        } else {
#pragma unroll
            for (int s = 0; s < SPW; ++s) skinny_store4(p, row, (strip0 + s) * FS + lq * 4, v[s] + resv[s][m], true);
        }
    }
    QTTS_TS_DRAINED(5);                    // stores acknowledged
    QTTS_TS_END(skinny, 0, p.K, p.N);
}

// ------------------------------------------------------------------------------------------ bf16, batch 17..32, deep K: split-K (round 6)
// At batch 17..32 (two 16-row tiles of x) a workgroup of skinny2_kernel pulls M x K x 2 B of x whatever its strip width: the talker's
// down-projection (K = 6144, 256 workgroups of ONE 8-feature strip) moves 393 KB of x against 98 KB of weights per workgroup -- 100 MB
// of L2 -> CU traffic for a 25 MB operator, 14.5 us per launch against 5.7 us at batch 8 (profiles/r03_skinny_b32_straight.md).  The x
// traffic of a launch is (N / features per workgroup) x M x K x 2 B: only WIDER workgroups cut it, and then too few of them are left to
// pull the operator.  So: a workgroup owns 32 features (SPW strips of the packing's FS) and 1 / KS of K -- the same 256 workgroups,
// a quarter (an eighth) of the x bytes each -- and the KS partial sums of a strip group are combined inside the launch:
//   * workgroups ks < KS - 1 store their sums as tagged 8-byte granules {fp32, tag} (write-through, granule.h) and leave;
//   * the group's LAST workgroup (ks = KS - 1, the highest block index of the group) reads them until they carry the launch's tag, adds
//     p0 + p1 + ... + its own in k order (fixed: run-to-run identical), + bias + residual, and writes the rows (fp32 + bf16 shadow).
// Tag = frame serial << 8 | launch slot (the engine numbers the split launches of a frame; a granule left by an earlier launch has
// another tag).  Producers never wait and have lower block indices than their reducer -- workgroups are dispatched in index order -- so a
// reducer's producers are running or done whenever it polls: no residency account is needed (unlike the all-to-all hand-offs of
// cp_attn_o / cp_mlp).  A reducer that never sees its tags (GRANULE_SPIN_LIMIT re-reads) raises the engine's flag and latches the
// generation's stop flag, as the fused launches do.
// No RMSNorm (o- and down-projections have none), no SwiGLU.  Every request of the launch is issued at entry (straight-line, TPW k-tiles
// per wave); wave w of the workgroup combines and publishes / reduces (strip, m-tile) pair w, w + NW, ...
// (tstamp build: thread 0 of the first workgroup -- a producer -- and of the last -- a reducer, blk bit 16 -- append a record of kind 7)
#if QTTS_TSTAMP
#define QTTS_KS_TS_END(red_)                                                                                                 \
    do {                                                                                                                     \
        if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) {                                          \
            const unsigned i_ = atomicAdd(&qtts::ts_cnt_skinny, 1u);                                                         \
            if (i_ < qtts::TS_CAP) {                                                                                         \
                qtts::TsRec r_;                                                                                              \
                for (int k_ = 0; k_ < 6; ++k_) r_.t[k_] = ts_[k_];                                                           \
                r_.kind = 7; r_.a = p.K; r_.b = p.N; r_.blk = (int)blockIdx.x | ((red_) << 16);                              \
                qtts::ts_log_skinny[i_] = r_;                                                                                \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
#else
#define QTTS_KS_TS_END(red_) do { } while (0)
#endif
template <int SPW, int FS, int NW, int TPW, int KS>
__global__ __launch_bounds__(NW * 64) void skinny2_ks_kernel(SkinnyParams p) {
    constexpr int MT = 2, KT = 32, NP = SPW * MT, PPW = (NP + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sk[];
    f32x4* red = reinterpret_cast<f32x4*>(smem_sk);             // [NW][NP][64]
    QTTS_TS_BEGIN();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int sg = blockIdx.x / KS, ks = blockIdx.x - sg * KS;
    const int nkt = p.K / KT;
    const int kt0 = ks * (NW * TPW) + wave;                      // this wave's tiles: kt0 + NW i
    const int strip0 = sg * SPW;
    const bool wlane = lj < FS;
    const bool reducer = ks == KS - 1;

    u32x4 wR[TPW][SPW], xR[TPW][MT];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) wR[i][s] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int m = 0; m < MT; ++m) xR[i][m] = (u32x4){0u, 0u, 0u, 0u};
    }
    if (FS == 16 || wlane) {                                     // one exec mask for the weight requests (lanes that own a weight row)
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int s = 0; s < SPW; ++s)
                wR[i][s] = skinny_wload(reinterpret_cast<const u32x4*>(p.Wp) + ((size_t)(strip0 + s) * nkt + kt0 + NW * i) * (FS * 4) + lq * FS + (wlane ? lj : 0));
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
        if (m * 16 + lj < p.M) {                                 // ... and one per m-tile for its x requests
            const unsigned short* xp = reinterpret_cast<const unsigned short*>(p.x) + (size_t)(m * 16 + lj) * p.ldx + lq * 8;
#pragma unroll
            for (int i = 0; i < TPW; ++i) xR[i][m] = *reinterpret_cast<const u32x4*>(xp + (size_t)(kt0 + NW * i) * KT);
        }
    // the reducer's epilogue operands, under the weight stream
    f32x4 resv[PPW], biasv[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        resv[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; biasv[j] = resv[j];
        const int pr = wave + NW * j, s = pr / MT, m = pr - s * MT;
        const int row = m * 16 + lj, col = (strip0 + s) * FS + lq * 4;
        if (reducer && pr < NP && row < p.M && lq * 4 < FS) {
            if (p.res) resv[j] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
            if (p.bias) biasv[j] = *reinterpret_cast<const f32x4*>(p.bias + col);
        }
    }
    const unsigned serial = p.ks_serial ? (unsigned)*p.ks_serial : 1u;
    const int done = p.done_flag ? *p.done_flag : 0;
    QTTS_TS(1);
    if (done) return;
    QTTS_TS_DRAINED(2);
    const unsigned tag = (serial << 8) | (unsigned)p.ks_slot;

    f32x4 acc[SPW][MT];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            bf16x8 xb;
            *reinterpret_cast<u32x4*>(&xb) = xR[i][m];
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                bf16x8 wa;
                *reinterpret_cast<u32x4*>(&wa) = wR[i][s];
                acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[s][m], 0, 0, 0);
            }
        }
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) red[(wave * NP + s * MT + m) * 64 + lane] = acc[s][m];
    QTTS_TS_DRAINED(3);
    __syncthreads();

    const WtBuf part = wt_buf(p.ks_part, p.ks_part_bytes);
    f32x4 own[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pr = wave + NW * j;
        own[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (pr < NP) {
            f32x4 t = red[(0 * NP + pr) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) t += red[(w2 * NP + pr) * 64 + lane];     // fixed order
            own[j] = t;
            if (!reducer) {
                const int off = (((int)blockIdx.x * NP + pr) * 64 + lane) * 32;
                wt_store16(part, off, (cu32x4){__float_as_uint(t[0]), tag, __float_as_uint(t[1]), tag});
                wt_store16(part, off + 16, (cu32x4){__float_as_uint(t[2]), tag, __float_as_uint(t[3]), tag});
            }
        }
    }
    if (!reducer) { QTTS_TS_DRAINED(4); QTTS_KS_TS_END(0); return; }
    // ---- the group's last workgroup: the other KS - 1 partial sums, in k order
    wt_first_pause(p.ks_pause);
    cu32x4 g[PPW][KS - 1][2];
    auto load_all = [&] {
#pragma unroll
        for (int j = 0; j < PPW; ++j)
#pragma unroll
            for (int k2 = 0; k2 < KS - 1; ++k2) {
                const int pr = wave + NW * j < NP ? wave + NW * j : 0;
                const int off = (((sg * KS + k2) * NP + pr) * 64 + lane) * 32;
                g[j][k2][0] = wt_load16(part, off);
                g[j][k2][1] = wt_load16(part, off + 16);
            }
    };
    auto all_fresh = [&]() -> bool {
        bool f = true;
#pragma unroll
        for (int j = 0; j < PPW; ++j)
#pragma unroll
            for (int k2 = 0; k2 < KS - 1; ++k2)
#pragma unroll
                for (int h = 0; h < 2; ++h) f = f && g[j][k2][h][1] == tag && g[j][k2][h][3] == tag;
        return __ballot(!f) == 0;                                  // (a wave re-reads together: one request stream)
    };
    load_all();
    for (int spins = 0; !all_fresh(); ++spins) {
        if (spins > GRANULE_SPIN_LIMIT) {
            if (p.ks_err) __hip_atomic_store(p.ks_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.ks_latch) __hip_atomic_store(p.ks_latch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        // (the re-read must stay IN the loop: raw buffer loads are read-only intrinsics, and a loop without a store lets the compiler hoist them --
        // the first build polled once and span on the stale registers; the clobber and the sleep's side effect pin them here)
        asm volatile("" ::: "memory");
        wt_first_pause(1);
        load_all();
    }
    QTTS_TS(4);
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int pr = wave + NW * j, s = pr / MT, m = pr - s * MT;
        const int row = m * 16 + lj, col = (strip0 + s) * FS + lq * 4;
        if (pr >= NP || row >= p.M || lq * 4 >= FS) continue;
        f32x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = __uint_as_float(g[j][0][r >> 1][(r & 1) * 2]);
#pragma unroll
        for (int k2 = 1; k2 < KS - 1; ++k2)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] += __uint_as_float(g[j][k2][r >> 1][(r & 1) * 2]);
        t += own[j];
        skinny_store4(p, row, col, t + biasv[j] + resv[j], true);
    }
    QTTS_TS_DRAINED(5);
    QTTS_KS_TS_END(1);
}

// ------------------------------------------------------------------------------------------ bf16, batch <= 8 (the benchmarked shape)
// The frame step at batch <= 8 (round 2, after the in-kernel timestamps of profiles/r02_tstamp_frame.md): the same arithmetic as
// skinny2_kernel<1, SPW, 8, FS, true, ..., EXACT> with every request unconditional and twice as wide.
//   * A wave owns PAIRS of adjacent k-tiles (pair j = wave + 8 i).  At batch <= 8 the MFMA's batch columns 8..15 are padding, so
//     the x operand of a pair is ONE request of all 64 lanes for 8 rows x 128 B -- whole cache lines -- instead of two
//     requests of 32 lanes for 8 rows x 64 B: lane (lj, lq) reads row lj & 7, 16-B piece lq + 4 (lj >> 3).  Lanes lj < 8 then hold
//     the B fragment of the even tile in place; the odd tile's fragment sits in the padding columns lj >= 8 and a DPP row
//     rotation by 8 brings it to the columns lj < 8.  Narrow strips (FS = 8: feature rows 8..15 of the A operand are padding)
//     fetch their weights the same way: one 1-KiB request per pair and strip, rotated for the odd tile.  What a rotation leaves in
//     the padding rows / columns only reaches output features >= FS or batch rows >= 8, which nobody stores.
//   * No exec mask and no conditional load anywhere: a conditional load makes the compiler merge the loaded value with the
//     register's previous value, and where the register allocator turned that merge into a copy it waited for the load --
//     `s_waitcnt vmcnt(0)` in the middle of the request burst, one full memory round trip (seen in several instantiations of
//     skinny2_kernel, different ones after every edit).  Rows >= M re-read row 0, absent residual / bias operands read x; the
//     epilogue selects.
// NP = pairs per wave (K = 512 NP), NORM: RMSNorm folded in (acc_ss rides on the matrix pipe as in skinny2_kernel).
__device__ inline u32x4 dpp_ror8(const u32x4& v) {
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v[e], 0x128 /* row_ror:8 */, 0xf, 0xf, false);
    return r;
}

// Kernel arguments (round 3, measured -2.7 % per frame, profiles/r03_ab_kpre.md): a by-value struct argument is never preloaded into
// SGPRs by the compiler's kernarg-preload feature (`-mllvm -amdgpu-kernarg-preload-count=16`, build.py FLAGS: the first 14 dwords
// of SCALAR arguments arrive in user SGPRs with the wave -- no `s_load` round trip in front of the first address computation).
// So the operands every request of the launch depends on travel as leading scalar arguments (5 pointers + 4 ints = 14 dwords);
// the struct follows and is only read by the epilogue.
#define QTTS_SK8_ARGS(p) (p).Wp, (p).x, (p).done_flag, (p).res, (p).bias, (p).ldx, (p).M, (p).K, (p).ldr, (p)
template <int SPW, int FS, int NP, bool NORM, int NW = 8>
__global__ __launch_bounds__(NW * 64) void skinny8_kernel(const void* kWp, const float* kx, const int* kdone, const float* kres, const float* kbias,
                                                          int kldx, int kM, int kK, int kldr, SkinnyParams p) {
    p.Wp = kWp; p.x = kx; p.done_flag = kdone; p.res = kres; p.bias = kbias; p.ldx = kldx; p.M = kM; p.K = kK; p.ldr = kldr;
    static_assert(FS == 16 || (FS == 8 && SPW == 1), "skinny8: strips of 16 features, or single strips of 8");
    constexpr int NS = SPW + 1;
    constexpr int WPP = FS == 16 ? 2 : 1;                        // weight requests per pair and strip (1 KiB each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sk[];
    f32x4* red = reinterpret_cast<f32x4*>(smem_sk);              // [NW][NS][64]
    QTTS_TS_BEGIN();

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int nkt = p.K >> 5;
    const int strip0 = blockIdx.x * SPW;

    // 16-B units; a pair of tiles is 2 x FS x 4 units
    const u32x4* wb[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
        wb[s] = reinterpret_cast<const u32x4*>(p.Wp) + (size_t)(strip0 + s) * nkt * (FS * 4)
                + (FS == 16 ? lq * 16 + lj : (lj >> 3) * 32 + lq * 8 + (lj & 7)) + (size_t)wave * (FS * 8);
    const unsigned short* xb = reinterpret_cast<const unsigned short*>(p.x) + (size_t)((lj & 7) < p.M ? (lj & 7) : 0) * p.ldx
                               + (lq + 4 * (lj >> 3)) * 8 + wave * 64;

    // ---- 1. every request of the launch, back to back: pair i of this wave = tiles 2 (wave + 8 i), + 1
    u32x4 wR[NP][SPW][WPP], xR[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
#pragma unroll
        for (int s = 0; s < SPW; ++s)
#pragma unroll
            for (int h = 0; h < WPP; ++h) wR[i][s][h] = skinny_wload(wb[s] + (size_t)i * (NW * FS * 8) + h * 64);
        xR[i] = *reinterpret_cast<const u32x4*>(xb + i * (NW * 64));
    }
    // epilogue operands (used by wave 0 only; requested by every wave so that no branch surrounds a load)
    const int colq = lq * 4 < FS ? lq * 4 : 0;
    const int rowc = lj < p.M ? lj : 0;
    f32x4 resv[SPW], biasv[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int col = p.act == ACT_SWIGLU8 ? blockIdx.x * 8 + (lq & 1) * 4
                                             : (p.act == ACT_SWIGLU ? blockIdx.x * 16 : (strip0 + s) * FS) + colq;
        const float* bp = p.bias ? p.bias + (strip0 + s) * FS + colq : reinterpret_cast<const float*>(p.x);
        const float* rp = p.res ? p.res + (size_t)rowc * p.ldr + col : reinterpret_cast<const float*>(p.x);
        biasv[s] = *reinterpret_cast<const f32x4*>(bp);
        resv[s] = *reinterpret_cast<const f32x4*>(rp);
    }
    const int done = p.done_flag ? *p.done_flag : 0;
    QTTS_TS(1);
    if (done) return;
    QTTS_TS_DRAINED(2);

    // ---- 2. consume in request order
    f32x4 acc[SPW], acc_ss = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < SPW; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        bf16x8 xe, xo;
        *reinterpret_cast<u32x4*>(&xe) = xR[i];
        *reinterpret_cast<u32x4*>(&xo) = dpp_ror8(xR[i]);
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            bf16x8 we, wo;
            *reinterpret_cast<u32x4*>(&we) = wR[i][s][0];
            if constexpr (FS == 16) *reinterpret_cast<u32x4*>(&wo) = wR[i][s][WPP - 1];
            else *reinterpret_cast<u32x4*>(&wo) = dpp_ror8(wR[i][s][0]);
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(we, xe, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo, xo, acc[s], 0, 0, 0);
        }
        if constexpr (NORM) {
            acc_ss = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xe, xe, acc_ss, 0, 0, 0);
            acc_ss = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xo, xo, acc_ss, 0, 0, 0);
        }
    }

    // ---- 3. cross-wave combine (fixed order) and epilogue by wave 0: the kernel's only barrier
#pragma unroll
    for (int s = 0; s < SPW; ++s) red[(wave * NS + s) * 64 + lane] = acc[s];
    if constexpr (NORM) red[(wave * NS + SPW) * 64 + lane] = acc_ss;
    QTTS_TS_DRAINED(3);
    __syncthreads();
    QTTS_TS(4);
    if (wave != 0) return;

    float rstd = 1.f;
    if constexpr (NORM) {        // diagonal of X.X^T: row lj sits in lane (lj, lj >> 2), component lj & 3
        const float* rf = reinterpret_cast<const float*>(red);
        const int src = (((lj >> 2) * 16 + lj) << 2) + (lj & 3);
        float ssum = rf[((0 * NS + SPW) * 64) * 4 + src];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) ssum += rf[((w2 * NS + SPW) * 64) * 4 + src];
        rstd = rsqrtf(ssum / (float)p.K + p.eps);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        f32x4 t = red[(0 * NS + s) * 64 + lane];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) t += red[(w2 * NS + s) * 64 + lane];
        v[s] = t * rstd + (p.bias ? biasv[s] : zero4);
    }
    if constexpr (SPW == 1 && FS == 16) {
        if (p.act == ACT_SWIGLU8) {      // rows 0..7 of the strip = gate, rows 8..15 = up of output columns 8 b .. 8 b + 7: the up quad of
            f32x4 up;                    // lane (lj, lq < 2) sits in lane (lj, lq + 2) = lane + 32
#pragma unroll
            for (int r = 0; r < 4; ++r) up[r] = __shfl(v[0][r], (lane + 32) & 63);
            if (lj < p.M && lq < 2) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (v[0][r] / (1.f + expf(-v[0][r]))) * up[r];
                o += p.res ? resv[0] : zero4;
                skinny_store4(p, lj, blockIdx.x * 8 + lq * 4, o, false);
            }
            QTTS_TS_DRAINED(5);
            QTTS_TS_END(skinny, 0, p.K, p.N);
            return;
        }
    }
    if (lj < p.M && lq * 4 < FS) {
        if (p.act == ACT_SWIGLU) {
            if constexpr (SPW == 2) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (v[0][r] / (1.f + expf(-v[0][r]))) * v[1][r];
                o += p.res ? resv[0] : zero4;
                skinny_store4(p, lj, blockIdx.x * 16 + lq * 4, o, false);
            }
        } else {
#pragma unroll
            for (int s = 0; s < SPW; ++s) skinny_store4(p, lj, (strip0 + s) * FS + lq * 4, v[s] + (p.res ? resv[s] : zero4), true);
        }
    }
    QTTS_TS_DRAINED(5);
    QTTS_TS_END(skinny, 0, p.K, p.N);
}

// ------------------------------------------------------------------------------------------ fp32 (exact parity mode)
// x fragments are loaded per k-tile from global/L2 and the row sums of squares come from `ss_in` (row_ss_kernel): this is the
// arithmetic of rounds 1-3.  Since round 4 the frame step at batch <= 8 runs skinny8_f32_kernel (below) instead; this kernel serves batch
// 9..64, odd K and the finalize-time tabulation of the code predictor's layer-0 rows.
template <int MT, int SPW, int NW>
__global__ __launch_bounds__(NW * 64) void skinny_f32_kernel(SkinnyParams p) {
    constexpr int KT = 16, FS = 16;
    constexpr int U = 8 / SPW;             // k-tiles per chunk; two chunks (16 x 1 KiB) in flight per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sk[];
    f32x4* red = reinterpret_cast<f32x4*>(smem_sk);          // [NW*SPW*MT*64]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lj = lane & 15, lq = lane >> 4;
    const int nkt = p.K / KT;
    const int my_tiles = (nkt - wave + NW - 1) / NW;       // tile = wave + NW*i
    const int nchunks = (my_tiles + U - 1) / U;
    const int strip0 = blockIdx.x * SPW;

    f32x4 acc[SPW][MT];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const u32x4* wbase[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
        wbase[s] = reinterpret_cast<const u32x4*>(p.Wp) + ((size_t)(strip0 + s) * nkt) * (FS * 4) + lq * FS + lj;

    auto load_chunk = [&](u32x4 (&w)[SPW][U], int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = c * U + u;
            const int kt = wave + NW * i;
#pragma unroll
            for (int s = 0; s < SPW; ++s)
                w[s][u] = (i < my_tiles && !QTTS_ABL(p, 8)) ? skinny_wload(wbase[s] + (size_t)kt * (FS * 4)) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    u32x4 wA[SPW][U], wB[SPW][U];
    load_chunk(wA, 0);
    load_chunk(wB, 1);

    f32x4 resv[SPW][MT], biasv[SPW];
    const bool epi_loads = wave == 0 && !QTTS_ABL(p, 4);
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        const int col = (p.act == ACT_SWIGLU ? blockIdx.x * 16 : (strip0 + s) * FS) + lq * 4;
        biasv[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (epi_loads && p.bias) biasv[s] = *reinterpret_cast<const f32x4*>(p.bias + (strip0 + s) * FS + lq * 4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            resv[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = m * 16 + lj;
            if (epi_loads && p.res && row < p.M && (p.act != ACT_SWIGLU || s == 0))
                resv[s][m] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
        }
    }
    const int done = (p.done_flag && !QTTS_ABL(p, 1)) ? *p.done_flag : 0;
    float rstd_l[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        rstd_l[m] = 1.f;
        const int row = m * 16 + lj;
        if (p.norm && row < p.M) rstd_l[m] = rsqrtf(p.ss_in[row] / (float)p.K + p.eps);
    }
    if (done) return;

    auto compute_chunk = [&](u32x4 (&w)[SPW][U], int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = c * U + u;
            if (i >= my_tiles) break;
            const int k = (wave + NW * i) * KT + lq * 4;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m * 16 + lj;
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < p.M && !QTTS_ABL(p, 2)) xv = *reinterpret_cast<const float4*>(p.x + (size_t)row * p.ldx + k);
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    f32x4 wa;
                    *reinterpret_cast<u32x4*>(&wa) = w[s][u];
                    acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0], xv.x, acc[s][m], 0, 0, 0);
                    acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1], xv.y, acc[s][m], 0, 0, 0);
                    acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2], xv.z, acc[s][m], 0, 0, 0);
                    acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[3], xv.w, acc[s][m], 0, 0, 0);
                }
            }
        }
    };
    for (int c = 0; c < nchunks; c += 2) {
        compute_chunk(wA, c);
        if (c + 2 < nchunks) load_chunk(wA, c + 2);
        if (c + 1 < nchunks) compute_chunk(wB, c + 1);
        if (c + 3 < nchunks) load_chunk(wB, c + 3);
    }
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) red[((wave * SPW + s) * MT + m) * 64 + lane] = acc[s][m];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + lj;
        f32x4 v[SPW];
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            f32x4 t = red[((0 * SPW + s) * MT + m) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) t += red[((w2 * SPW + s) * MT + m) * 64 + lane];
            v[s] = t * rstd_l[m] + biasv[s];
        }
        if (row >= p.M) continue;
        if (p.act == ACT_SWIGLU) {
            if constexpr (SPW == 2) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (v[0][r] / (1.f + expf(-v[0][r]))) * v[1][r];
                o += resv[0][m];
                *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + blockIdx.x * 16 + lq * 4) = o;
            }
        } else {
#pragma unroll
            for (int s = 0; s < SPW; ++s)
                *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + (strip0 + s) * FS + lq * 4) = v[s] + resv[s][m];
        }
    }
}

// ------------------------------------------------------------------------------------------ fp32, batch <= 8 (round 4)
// The parity mode's frame-step kernel: skinny8_kernel's design carried over to the exact-fp32 arithmetic (v_mfma_f32_16x16x4_f32
// chains, fp32 weights and activations).  Round 3 left the fp32 GEMM as rounds 1-2 had it -- x fragments loaded tile by tile inside
// the MFMA loop (a dependent L2 round trip per k-tile), guarded weight requests, and one extra `row_ss_kernel` launch in front of
// every normalised GEMM: 7.03 ms per frame for 10.3 GB of weights = 0.18 of the HBM peak, below the bf16 kernel's 0.29.
//   * A wave owns PAIRS of adjacent 16-wide k-tiles (pair j = wave + NW i).  At batch <= 8 the MFMA's batch columns 8..15 are
//     padding, so the x operand of a pair is ONE request of all 64 lanes for 8 rows x 128 B (whole cache lines): lane (lj, lq) reads
//     row lj & 7, floats 32 j + 16 (lj >> 3) + 4 lq .. + 4.  Lanes lj < 8 hold the even tile's B fragments in place, a DPP row rotation
//     by 8 brings the odd tile's out of the padding columns.
//   * Every request of a chunk is unconditional and issued back to back (rows >= M re-read row 0, absent residual / bias operands read
//     x; the epilogue selects).  CH = 1 | 2: everything is requested at kernel entry; CH = 3 (K = 6144: 24 pairs per wave would need
//     288 operand registers): two chunks at entry, the third into the first chunk's registers once that one is consumed.
//   * The RMSNorm row sum of squares is taken from the x fragments the wave holds anyway (4 FMAs per pair and lane, reduced over the
//     row's 8 lanes by DPP / bpermute and over the waves in fixed order at the combine): the 206 `row_ss_kernel` launches of an fp32
//     frame step are gone.
//   * Leading scalar arguments arrive preloaded in SGPRs as for skinny8_kernel.
// Summation order: a wave accumulates its pairs in ascending k, 4 MFMAs per tile (e = 0..3: k = 4 lq + e); waves are combined in
// ascending order.  This is not the order of skinny_f32_kernel (tiles dealt singly to the waves), so results differ in the last bits;
// the fp32 goldens (greedy codes, all 16 codebooks) are the acceptance test, on the emulator and on the MI355X.
__device__ inline f32x4 dpp_ror8_f(const f32x4& v) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[e]), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    return r;
}
// Split-K with the combine in the CONSUMER's prologue (round 4, second step).  The N = 1024 / 2048 operators (o-proj, down-proj) are 64 /
// 128 strips of 16 features: 64 workgroups pulling 128-196 KB each ran at 1.1-1.4 TB/s (profiles/r04_f32_skinny8.md), and narrower
// strips would spend 3/4 of an fp32 16x16x4 MFMA on padding.  Instead:
//   KS = 2  (producer): workgroup (strip, half) accumulates its half of K and writes its partial sums to `out + half * part_stride`
//           -- no bias, no activation, no normalisation; half 0 adds the residual to its sums (8 x 16 floats it reads anyway), half 1
//           writes raw sums; twice the workgroups, half the chain each;
//   COMB    (consumer: the NEXT decode GEMM, which reads that output as its x anyway): every x fragment is formed as
//           xp[0] + xp[1] = (residual + half 0) + half 1 from two requests instead of one -- the launch-boundary reduce costs no launch
//           and no barrier -- and workgroup 0 writes the combined rows to `x_out`, which is the residual stream from here on.  The
//           RMSNorm statistics come from the combined fragments.  (First version of this round: the consumer read residual, half 0 and
//           half 1 -- three requests; moving the residual into half 0 is the same sum in the same order, bit for bit, for two thirds of
//           the consumer's x traffic.)
// Fixed summation order ((residual + half 0) + half 1), so results are run-to-run identical; they differ in the last bits from the
// unsplit kernel -- the fp32 goldens are the acceptance test.
template <int SPW, int NP, int CH, bool NORM, int NW, int KS = 1, bool COMB = false>
__global__ __launch_bounds__(NW * 64) void skinny8_f32_kernel(const void* kWp, const float* kx, const int* kdone, const float* kres, const float* kbias,
                                                              int kldx, int kM, int kK, int kldr, SkinnyParams p) {
    p.Wp = kWp; p.x = kx; p.done_flag = kdone; p.res = kres; p.bias = kbias; p.ldx = kldx; p.M = kM; p.K = kK; p.ldr = kldr;
    static_assert(CH >= 1 && CH <= 3, "skinny8_f32: 1..3 chunks");
    static_assert(KS == 1 || (KS == 2 && !NORM && !COMB && SPW == 1), "skinny8_f32: the split-K producer is a plain single-strip GEMM");
    constexpr int NSET = CH < 2 ? CH : 2;                        // register sets
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sk[];
    f32x4* red = reinterpret_cast<f32x4*>(smem_sk);              // [NW][SPW][64]
    float* ssl = reinterpret_cast<float*>(red + NW * SPW * 64);  // [NW][16]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int nkt = p.K >> 4;                                    // k-tiles of the whole operator (a strip's stride)
    const int bidx = KS == 1 ? blockIdx.x : blockIdx.x / KS;     // strip (pair) index
    const int kh = KS == 1 ? 0 : blockIdx.x % KS;                // which half of K
    const int strip0 = bidx * SPW;

    // 16-B units; a tile is 64 units, a pair 128
    const u32x4* wb[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
        wb[s] = reinterpret_cast<const u32x4*>(p.Wp) + (size_t)(strip0 + s) * nkt * 64 + (size_t)kh * (nkt / KS) * 64 + lq * 16 + lj + (size_t)wave * 128;
    const size_t xoff = (size_t)((lj & 7) < p.M ? (lj & 7) : 0) * p.ldx + kh * (p.K / KS) + (lj >> 3) * 16 + lq * 4 + wave * 32;
    const float* xb = p.x + xoff;

    u32x4 wR[NSET][NP][SPW][2];
    f32x4 xR[NSET][NP], pR[COMB ? NSET : 1][COMB ? NP : 1];
    auto request = [&](int set, int c) {                          // chunk c = pairs wave + NW (c NP + i)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
#pragma unroll
            for (int s = 0; s < SPW; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h) wR[set][i][s][h] = skinny_wload(wb[s] + (size_t)(c * NP + i) * (NW * 128) + h * 64);
            if constexpr (COMB) {
                xR[set][i] = *reinterpret_cast<const f32x4*>(p.xp + xoff + (c * NP + i) * (NW * 32));                      // residual + half 0
                pR[set][i] = *reinterpret_cast<const f32x4*>(p.xp + p.xp_stride + xoff + (c * NP + i) * (NW * 32));       // half 1
            } else xR[set][i] = *reinterpret_cast<const f32x4*>(xb + (c * NP + i) * (NW * 32));
        }
    };
    // ---- 1. the requests, back to back
    request(0, 0);
    if constexpr (CH >= 2) request(1, 1);
    // epilogue operands (used by wave 0 only; requested by every wave so that no branch surrounds a load)
    const int rowc = lj < p.M ? lj : 0;
    constexpr bool EPI = KS == 1 && !COMB;                       // (a combining consumer takes neither bias nor residual; a split-K producer no bias)
    constexpr bool EPI_RES = !COMB;
    f32x4 resv[SPW], biasv[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        biasv[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
        resv[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int col = (p.act == ACT_SWIGLU ? bidx * 16 : (strip0 + s) * 16) + lq * 4;
        if constexpr (EPI) {
            const float* bp = p.bias ? p.bias + (strip0 + s) * 16 + lq * 4 : p.x;
            biasv[s] = *reinterpret_cast<const f32x4*>(bp);
        }
        if constexpr (EPI_RES) {
            const float* rp = p.res ? p.res + (size_t)rowc * p.ldr + col : p.x;
            resv[s] = *reinterpret_cast<const f32x4*>(rp);
        }
    }
    const int done = p.done_flag ? *p.done_flag : 0;
    if (done) return;

    // ---- 2. consume in request order
    f32x4 acc[SPW];
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < SPW; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto consume = [&](int set, int c) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            f32x4 xe = xR[set][i];
            if constexpr (COMB) {
                xe = xe + pR[set][i];                             // the launch-boundary reduce: (residual + half 0) + half 1
                if (blockIdx.x == 0 && (lj & 7) < p.M)            // one workgroup writes the combined rows: the residual stream from here on
                    *reinterpret_cast<f32x4*>(p.x_out + xoff + (c * NP + i) * (NW * 32)) = xe;
            }
            const f32x4 xo = dpp_ror8_f(xe);
            if constexpr (NORM) {                                 // this lane's 4 values belong to row lj & 7 (even or odd tile of the pair)
#pragma unroll
                for (int e = 0; e < 4; ++e) ss = fmaf(xe[e], xe[e], ss);
            }
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                f32x4 we, wo;
                *reinterpret_cast<u32x4*>(&we) = wR[set][i][s][0];
                *reinterpret_cast<u32x4*>(&wo) = wR[set][i][s][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(we[e], xe[e], acc[s], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wo[e], xo[e], acc[s], 0, 0, 0);
            }
        }
    };
    consume(0, 0);
    if constexpr (CH == 3) request(0, 2);
    if constexpr (CH >= 2) consume(1, 1);
    if constexpr (CH == 3) consume(0, 2);

    // ---- 3. cross-wave combine (fixed order) and epilogue by wave 0: the kernel's only barrier
#pragma unroll
    for (int s = 0; s < SPW; ++s) red[(wave * SPW + s) * 64 + lane] = acc[s];
    if constexpr (NORM) {        // row lj & 7: lanes (lj & 7, lj & 7 | 8) x lq = 0..3 of this wave
        float t = ss + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ss), 0x128, 0xf, 0xf, false));
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        if (lane < 16) ssl[wave * 16 + lane] = t;
    }
    __syncthreads();
    if (wave != 0) return;

    float rstd = 1.f;
    if constexpr (NORM) {
        float ssum = ssl[lj & 7];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) ssum += ssl[w2 * 16 + (lj & 7)];
        rstd = rsqrtf(ssum / (float)p.K + p.eps);
    }
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        f32x4 t = red[(0 * SPW + s) * 64 + lane];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) t += red[(w2 * SPW + s) * 64 + lane];
        v[s] = t * rstd + (EPI && p.bias ? biasv[s] : zero4);
    }
    if (lj < p.M) {
        if constexpr (KS > 1) {              // this half's sums; half 0 carries the residual (the consumer adds the two halves)
            *reinterpret_cast<f32x4*>(p.out + (size_t)kh * p.part_stride + (size_t)lj * p.ldo + strip0 * 16 + lq * 4) =
                v[0] + (kh == 0 && p.res ? resv[0] : zero4);
        } else if (p.act == ACT_SWIGLU) {
            if constexpr (SPW == 2) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (v[0][r] / (1.f + expf(-v[0][r]))) * v[1][r];
                o += EPI && p.res ? resv[0] : zero4;
                *reinterpret_cast<f32x4*>(p.out + (size_t)lj * p.ldo + bidx * 16 + lq * 4) = o;
            }
        } else {
#pragma unroll
            for (int s = 0; s < SPW; ++s)
                *reinterpret_cast<f32x4*>(p.out + (size_t)lj * p.ldo + (strip0 + s) * 16 + lq * 4) = v[s] + (EPI && p.res ? resv[s] : zero4);
        }
    }
}

// bf16 mode: a GEMM input may arrive as the producer's bf16 copy (x_bf16) for any M <= 64 -- nothing is staged through LDS any
// more, so there is no capacity condition left (round 1: M <= 16 up to K = 7096, M <= 32 up to K = 2344).
bool skinny_takes_bf16_x(int M, int K, bool bf16) { return bf16 && M >= 1 && M <= 64 && K % 32 == 0; }
// ACT_SWIGLU8 lives in skinny8_kernel only: the K for which that kernel is instantiated (launch8_nw)
bool skinny_swiglu8_takes(int K) { return K == 1024 || K == 2048 || K == 3072 || K == 6144; }

template <int MT, int SPW, int NW, int FS, bool XB16, int U, bool EXACT, int NCH>
static void launch2_n(const SkinnyParams& p, hipStream_t st) {
    const int grid = p.N / (FS * SPW);
    const size_t lds = (size_t)NW * (SPW * MT + MT) * 64 * 16;
    QTTS_REQUIRE(lds <= 160 * 1024, QTTS_ERR_LIMIT, "skinny: LDS budget exceeded");
    auto kern = skinny2_kernel<MT, SPW, NW, FS, XB16, U, EXACT, NCH>;
    if (lds > 48 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
    QTTS_SK_LAUNCH(kern, dim3(grid), dim3(NW * 64), lds, st, p);
}
template <int MT, int SPW, int NW, int FS, bool XB16, int U, bool EXACT>
static void launch2_u(const SkinnyParams& p, int nchunks, hipStream_t st) {
    if constexpr (EXACT && MT == 1) {      // the frame step at batch <= 16: straight-line kernels for 1, 2 or 3 chunks
        if (nchunks == 1) { launch2_n<MT, SPW, NW, FS, XB16, U, true, 1>(p, st); return; }
        if (nchunks == 2) { launch2_n<MT, SPW, NW, FS, XB16, U, true, 2>(p, st); return; }
        if (nchunks == 3) { launch2_n<MT, SPW, NW, FS, XB16, U, true, 3>(p, st); return; }
    }
    // Round 3: the same for batch 17..32 (two 16-row tiles; chunks of 4 / SPW k-tiles, so K <= 3072 at 8 waves: every code-predictor
    // GEMM and the talker's q|k|v, o and gate|up).  In the generic loop the waves of a launch drift apart by a memory round trip per
    // chunk pair -- in-kernel timestamps at batch 32: 1.0-2.4 us between the first wave's last MFMA and the barrier
    // (profiles/r03_skinny_b32_straight.md; the A/B switch for the loop form was retired in round 6: measured in rounds 3 and 4).
    if constexpr (EXACT && MT == 2 && XB16 && NW == 8) {
        if (nchunks == 1) { launch2_n<MT, SPW, NW, FS, XB16, U, true, 1>(p, st); return; }
        if (nchunks == 2) { launch2_n<MT, SPW, NW, FS, XB16, U, true, 2>(p, st); return; }
        if (nchunks == 3) { launch2_n<MT, SPW, NW, FS, XB16, U, true, 3>(p, st); return; }
        if constexpr (SPW == 2) {      // (chunks of 2 k-tiles: the talker's gate|up, K = 2048, is 4 of them)
            if (nchunks == 4) { launch2_n<MT, SPW, NW, FS, XB16, U, true, 4>(p, st); return; }
        }
    }
    launch2_n<MT, SPW, NW, FS, XB16, U, EXACT, 0>(p, st);
}
template <int MT, int SPW, int NW, int FS, bool XB16>
static void launch2_x(const SkinnyParams& p, hipStream_t st) {
    constexpr int UMAX = (MT == 1 ? 8 : (MT == 2 ? 4 : 2)) / SPW;
    if constexpr (XB16 && NW == 8) {       // the frame step's shapes: K / 32 tiles dealt evenly to 8 waves
        const int nkt = p.K / 32;
        if (nkt % NW == 0) {
            const int tpw = nkt / NW;
            if (tpw % UMAX == 0) { launch2_u<MT, SPW, NW, FS, XB16, UMAX, true>(p, tpw / UMAX, st); return; }
            if constexpr (UMAX == 8) {
                if (tpw % 4 == 0) { launch2_u<MT, SPW, NW, FS, XB16, 4, true>(p, tpw / 4, st); return; }
            }
        }
    }
    launch2_u<MT, SPW, NW, FS, XB16, UMAX, false>(p, 0, st);
}
template <int MT, int SPW, int NW, int FS>
static void launch2_one(const SkinnyParams& p, hipStream_t st) {
    if (p.x_bf16) launch2_x<MT, SPW, NW, FS, true>(p, st); else launch2_x<MT, SPW, NW, FS, false>(p, st);
}
template <int MT, int NW>
static void launch2_mt(const SkinnyParams& p, int spw, int fs, hipStream_t st) {
    if (spw == 2) launch2_one<MT, 2, NW, 16>(p, st);
    else if (fs == 16) launch2_one<MT, 1, NW, 16>(p, st);
    else if (fs == 8) launch2_one<MT, 1, NW, 8>(p, st);
    else launch2_one<MT, 1, NW, 4>(p, st);
}

template <int MT, int SPW, int NW>
static void launch_f32_one(const SkinnyParams& p, hipStream_t st) {
    const int grid = p.N / (16 * SPW);
    const size_t lds = (size_t)NW * SPW * MT * 64 * 16;
    QTTS_SK_LAUNCH((skinny_f32_kernel<MT, SPW, NW>), dim3(grid), dim3(NW * 64), lds, st, p);
}
template <int MT>
static void launch_f32_mt(const SkinnyParams& p, int spw, int nw, hipStream_t st) {
    if (nw == 8) { if (spw == 2) launch_f32_one<MT, 2, 8>(p, st); else launch_f32_one<MT, 1, 8>(p, st); }
    else         { if (spw == 2) launch_f32_one<MT, 2, 4>(p, st); else launch_f32_one<MT, 1, 4>(p, st); }
}

// batch 17..32, no norm, plain epilogue: the split-K kernel (round 6).  Instantiations = the o- and down-projections of the released
// stacks under the engine's strip widths: 1 talker down 1.7B (N 2048, K 6144, fs 8), 2 down of the 1024-wide stacks (N 1024, K 3072,
// fs 8), 3 talker o 1.7B (N 2048, K 2048, fs 16), 4 o of the 1024-wide stacks (N 1024, K 2048, fs 8).  QTTS_SKINNY_KS=0: skinny2_kernel;
// QTTS_SKINNY_KS_MINK: the smallest K that splits.  Default 6144 -- the talker's down-projection only: 14.65 -> 10.7 us per launch, frame step at
// batch 32 4.46 -> 4.31 ms (-3.4 %).  At K = 3072 (the 1024-wide stacks' down-projection, 8 parts) the combine costs what the split saves
// (8.36 -> 8.9 us per launch: 4.39 ms), at K = 2048 it loses (4.55 ms): in-process A/B and timelines in profiles/r06_skinny_ksplit.md.
static int ksplit_choice(int M, int N, int K, int fs) {
    if (M <= 16 || M > 32 || N % 32 != 0 || K % 32 != 0) return 0;
    const int groups = N / 32, nkt = K / 32, ks = groups >= 64 ? 4 : 8;
    if (fs == 8 && ks == 4 && nkt == 4 * 8 * 6) return 1;
    if (fs == 8 && ks == 8 && nkt == 8 * 4 * 3) return 2;
    if (fs == 16 && ks == 4 && nkt == 4 * 8 * 2) return 3;
    if (fs == 8 && ks == 8 && nkt == 8 * 8 * 1) return 4;
    return 0;
}
bool skinny_ksplit_takes(int M, int N, int K, int fs) { return ksplit_choice(M, N, K, fs) != 0; }
template <int SPW, int FS, int NW, int TPW, int KS>
static void launch_ks_n(const SkinnyParams& p, hipStream_t st) {
    const int grid = p.N / (FS * SPW) * KS;
    const size_t lds = (size_t)NW * SPW * 2 * 64 * 16;
    QTTS_REQUIRE((size_t)grid * SPW * 2 * 64 * 32 <= p.ks_part_bytes, QTTS_ERR_LIMIT, "skinny: the split-K workspace is too small for this launch");
    QTTS_REQUIRE(p.ks_slot >= 0 && p.ks_slot < 256, QTTS_ERR_ARG, "skinny: split-K launch slot must be < 256");
    auto kern = skinny2_ks_kernel<SPW, FS, NW, TPW, KS>;
    if (lds > 48 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
    QTTS_SK_LAUNCH(kern, dim3(grid), dim3(NW * 64), lds, st, p);
}
static bool launch_skinny_ks(const SkinnyParams& p, int fs, hipStream_t st) {
    if (!p.ks_part || !p.x_bf16 || p.norm || p.act != ACT_NONE || p.out_bf16 || p.ksplit || QTTS_ABL(p, 15)) return false;
    if (!QTTS_OPT_ON("QTTS_SKINNY_KS") || p.K < QTTS_OPT_INT("QTTS_SKINNY_KS_MINK", 6144)) return false;
    switch (ksplit_choice(p.M, p.N, p.K, fs)) {
        case 1: launch_ks_n<4, 8, 8, 6, 4>(p, st); return true;
        case 2: launch_ks_n<4, 8, 4, 3, 8>(p, st); return true;
        case 3: launch_ks_n<2, 16, 8, 2, 4>(p, st); return true;
        case 4: launch_ks_n<4, 8, 8, 1, 8>(p, st); return true;
        default: return false;
    }
}

static int skinny_spw(int N, int fs, bool swiglu) {
    if (swiglu) return 2;
    return (fs == 16 && N / 16 >= 1024 && (N / 16) % 2 == 0) ? 2 : 1;
}

// batch <= 8, bf16 x, K a multiple of 512 (whole tile pairs for 8 waves): the frame step's kernel.  QTTS_SKINNY8=0 falls back to
// skinny2_kernel (A/B; QTTS_ENV: one process compares both through qtts_set_option).
template <int SPW, int FS, int NP, int NW = 8>
static void launch8_n(const SkinnyParams& p, hipStream_t st) {
    const int grid = p.N / (FS * SPW);
    const size_t lds = (size_t)NW * (SPW + 1) * 64 * 16;
    if (p.norm) QTTS_SK_LAUNCH((skinny8_kernel<SPW, FS, NP, true, NW>), dim3(grid), dim3(NW * 64), lds, st, QTTS_SK8_ARGS(p));
    else QTTS_SK_LAUNCH((skinny8_kernel<SPW, FS, NP, false, NW>), dim3(grid), dim3(NW * 64), lds, st, QTTS_SK8_ARGS(p));
}
// waves per workgroup: fewer waves = fewer partial sums to combine and a shorter barrier, more tile pairs (registers) per wave.
// Default 4 for matrices below 16 MB (in-process A/B, GPU call 14: 2.854 vs 2.884 ms per frame with 8, both repetitions), 8 above;
// QTTS_SKINNY8_NW = 8 | 4 | 2 | 1 asks for another count where that instantiation exists (A/B; QTTS_ENV).
template <int SPW, int FS, int NW>
static bool launch8_nw(const SkinnyParams& p, hipStream_t st) {
    constexpr int REGS_PER_PAIR = (FS == 16 ? 8 : 4) * SPW + 4;          // operand VGPRs per tile pair
    switch (p.K / 512) {
#define QTTS_S8_CASE(KQ)                                                                                         \
        case KQ:                                                                                                 \
            if constexpr ((KQ * 8) % NW == 0 && (KQ * 8 / NW) * REGS_PER_PAIR <= 200) {                           \
                launch8_n<SPW, FS, KQ * 8 / NW, NW>(p, st);                                                      \
                return true;                                                                                     \
            }                                                                                                    \
            return false;
        QTTS_S8_CASE(2) QTTS_S8_CASE(4) QTTS_S8_CASE(6) QTTS_S8_CASE(12)
#undef QTTS_S8_CASE
        default: return false;
    }
}
template <int SPW, int FS>
static bool launch8_fs(const SkinnyParams& p, hipStream_t st) {
    const char* e = QTTS_ENV("QTTS_SKINNY8_NW");
    // (matrices of 16 MB and more -- the talker's qkv, gate/up and down projections -- are bound by the stream itself and keep 8
    // waves: in-kernel timestamps of GPU call 18, 8 vs 4 waves: 3.83 vs 4.15, 8.5 vs 9.4, 5.95 vs 6.6 us)
    const int dflt = (size_t)p.N * p.K * 2 >= ((size_t)16 << 20) ? 8 : 4;
    const int want = (e && (e[0] == '8' || e[0] == '4' || e[0] == '2' || e[0] == '1')) ? e[0] - '0' : dflt;
    if (want <= 1 && launch8_nw<SPW, FS, 1>(p, st)) return true;
    if (want <= 2 && launch8_nw<SPW, FS, 2>(p, st)) return true;
    if (want <= 4 && launch8_nw<SPW, FS, 4>(p, st)) return true;
    return launch8_nw<SPW, FS, 8>(p, st);
}
static bool launch_skinny8(const SkinnyParams& p, int spw, int fs, hipStream_t st) {
    const char* e = QTTS_ENV("QTTS_SKINNY8");
    if (e && e[0] == '0') return false;
    if (!p.x_bf16 || p.M > 8 || p.K % 512 != 0 || QTTS_ABL(p, 15)) return false;
    if (fs == 16) return spw == 2 ? launch8_fs<2, 16>(p, st) : launch8_fs<1, 16>(p, st);
    if (fs == 8 && spw == 1) return launch8_fs<1, 8>(p, st);
    return false;
}

// fp32, batch <= 8 (skinny8_f32_kernel).  QTTS_SKINNY8F=0 keeps skinny_f32_kernel + row_ss_kernel (A/B; read once: the engine decides
// with the same function whether a normalised GEMM needs the row sums of squares from a launch of its own).
static bool skinny8f_enabled() {
    const bool on = QTTS_OPT_ON("QTTS_SKINNY8F");
    return on;
}
template <int SPW, int NP, int CH, int NW>
static void launch8f_n(const SkinnyParams& p, hipStream_t st) {
    const int grid = p.N / (16 * SPW);
    const size_t lds = (size_t)NW * SPW * 64 * 16 + (size_t)NW * 16 * 4;
    if (p.norm) QTTS_SK_LAUNCH((skinny8_f32_kernel<SPW, NP, CH, true, NW>), dim3(grid), dim3(NW * 64), lds, st, QTTS_SK8_ARGS(p));
    else QTTS_SK_LAUNCH((skinny8_f32_kernel<SPW, NP, CH, false, NW>), dim3(grid), dim3(NW * 64), lds, st, QTTS_SK8_ARGS(p));
}
// split-K producer (two workgroups per strip) and combining consumer (x = x + xp[0] + xp[1]; always a normalised GEMM: q|k|v, gate|up)
template <int NP, int NW>
static void launch8f_split(const SkinnyParams& p, hipStream_t st) {
    const size_t lds = (size_t)NW * 64 * 16 + (size_t)NW * 16 * 4;
    QTTS_SK_LAUNCH((skinny8_f32_kernel<1, NP, 1, false, NW, 2, false>), dim3(p.N / 16 * 2), dim3(NW * 64), lds, st, QTTS_SK8_ARGS(p));
}
template <int SPW, int NP, int NW>
static void launch8f_comb(const SkinnyParams& p, hipStream_t st) {
    const size_t lds = (size_t)NW * SPW * 64 * 16 + (size_t)NW * 16 * 4;
    QTTS_SK_LAUNCH((skinny8_f32_kernel<SPW, NP, 1, true, NW, 1, true>), dim3(p.N / (16 * SPW)), dim3(NW * 64), lds, st, QTTS_SK8_ARGS(p));
}
// waves per workgroup and chunks by K (pairs = K / 32): operand registers per pair = 8 SPW + 4, at most ~200 per register set pair.
//   K    pairs   SPW = 1                     SPW = 2 (SwiGLU)
//   1024   32    8 waves x 4                 8 waves x 4
//   2048   64    16 waves x 4                16 waves x 4
//   3072   96    8 waves x 12                8 waves x 3 chunks of 4
//   6144  192    8 waves x 3 chunks of 8     --
// (16 waves x 6 pairs was tried for K = 3072 / 6144: a 1024-thread workgroup caps a wave at 128 registers and the kernel spilled)
template <int SPW>
static bool launch8f_spw(const SkinnyParams& p, hipStream_t st) {
    const int nw_env = QTTS_OPT_INT("QTTS_SKINNY8F_NW", 0);      // (the emulator test walks the instantiations)
    if (p.ksplit == 2) {                   // producer: K / 2 per workgroup
        if constexpr (SPW == 1) {
            switch (p.K) {
                case 2048: launch8f_split<4, 8>(p, st); return true;
                case 3072: launch8f_split<6, 8>(p, st); return true;
                case 6144: launch8f_split<12, 8>(p, st); return true;
                default: return false;
            }
        }
        return false;
    }
    if (p.xp) {                            // consumer of a split producer
        switch (p.K) {
            case 1024: launch8f_comb<SPW, 4, 8>(p, st); return true;
            case 2048: launch8f_comb<SPW, 4, 16>(p, st); return true;
            default: return false;
        }
    }
    // QTTS_SKINNY8F_NW = 4 | 8 | 16 asks for another wave count where that instantiation exists (A/B)
    switch (p.K) {
        case 1024:
            if (nw_env == 4) launch8f_n<SPW, 8, 1, 4>(p, st);
            else if (nw_env == 16) launch8f_n<SPW, 2, 1, 16>(p, st);
            else launch8f_n<SPW, 4, 1, 8>(p, st);
            return true;
        case 2048:          // 16 waves x 4 pairs: 4.53 vs 4.71 ms per fp32 frame with 8 x 8 (GPU call 2 of round 4)
            if (nw_env == 8) launch8f_n<SPW, 8, 1, 8>(p, st); else launch8f_n<SPW, 4, 1, 16>(p, st);
            return true;
        case 3072:
            if constexpr (SPW == 1) { if (nw_env == 16) launch8f_n<1, 6, 1, 16>(p, st); else launch8f_n<1, 12, 1, 8>(p, st); }
            else launch8f_n<2, 4, 3, 8>(p, st);
            return true;
        case 6144:
            if constexpr (SPW == 1) { launch8f_n<1, 8, 3, 8>(p, st); return true; }
            return false;
        default: return false;
    }
}
// the K for which skinny8_f32_kernel is instantiated: a normalised fp32 GEMM with M <= 8 and such a K needs no ss_in
bool skinny_f32_inline_norm(int M, int K) {
    return skinny8f_enabled() && M >= 1 && M <= 8 && (K == 1024 || K == 2048 || K == 3072);
}
// split-K producer / combining consumer shapes (the engine asks before it plans a layer that way)
bool skinny_f32_splitk_takes(int M, int K_producer, int K_consumer) {
    const bool on = QTTS_OPT_ON("QTTS_SKINNY8F_SPLITK");     // (=0: A/B)
    return on && skinny8f_enabled() && M >= 1 && M <= 8 && (K_producer == 2048 || K_producer == 3072 || K_producer == 6144) &&
           (K_consumer == 1024 || K_consumer == 2048);
}
static bool launch_skinny8_f32(const SkinnyParams& p, int spw, hipStream_t st) {
    if (p.ksplit == 2 || p.xp) {           // no other kernel implements these: refuse loudly instead of falling through
        QTTS_REQUIRE(skinny8f_enabled() && p.M <= 8 && !p.x_bf16 && !p.out_bf16 && !p.out16, QTTS_ERR_ARG, "skinny: split-K / combine need the fp32 batch <= 8 kernel");
        QTTS_REQUIRE(!(p.ksplit == 2 && p.xp), QTTS_ERR_ARG, "skinny: a split-K producer cannot also combine");
        if (p.ksplit == 2) QTTS_REQUIRE(!p.norm && !p.bias && p.act == ACT_NONE && spw == 1, QTTS_ERR_ARG, "skinny: the split-K producer is a plain GEMM (+ residual)");
        if (p.xp) QTTS_REQUIRE(p.norm && p.x_out && p.x_out != p.xp && p.x_out != p.xp + p.xp_stride && !p.bias && !p.res, QTTS_ERR_ARG,
                               "skinny: the combining consumer is a normalised GEMM without bias / residual, writing x_out outside the halves");
        const bool ok = spw == 2 ? launch8f_spw<2>(p, st) : launch8f_spw<1>(p, st);
        QTTS_REQUIRE(ok, QTTS_ERR_ARG, "skinny: no split-K / combine instantiation for this K");
        return true;
    }
    if (!skinny8f_enabled() || p.M > 8 || p.x_bf16 || p.out_bf16 || p.out16) return false;
    if (spw == 2 && p.K == 6144) return false;
    if (!(p.K == 1024 || p.K == 2048 || p.K == 3072 || p.K == 6144)) return false;
    return spw == 2 ? launch8f_spw<2>(p, st) : launch8f_spw<1>(p, st);
}

void launch_skinny(const SkinnyParams& p, bool bf16, hipStream_t st) {
    const int KT = bf16 ? 32 : 16;
    const int fs = p.fs ? p.fs : 16;
    QTTS_REQUIRE(fs == 16 || fs == 8 || fs == 4, QTTS_ERR_ARG, "skinny: fs must be 16, 8 or 4");
    QTTS_REQUIRE(fs == 16 || bf16, QTTS_ERR_ARG, "skinny: narrow strips (fs < 16) are only built for the bf16 kernel");
    QTTS_REQUIRE(p.N % 16 == 0, QTTS_ERR_ARG, "skinny: N % 16");
    QTTS_REQUIRE(p.K % KT == 0, QTTS_ERR_ARG, "skinny: K must be a multiple of the k-tile");
    QTTS_REQUIRE(p.M >= 1 && p.M <= 64, QTTS_ERR_LIMIT, "skinny: 1 <= M <= 64");
#if !QTTS_ABLATE
    QTTS_REQUIRE(p.ablate == 0, QTTS_ERR_ARG, "skinny: perf-ablation flags need the `ablate` build variant (python qwen3-tts_amd/build.py --variant ablate)");
#endif
    QTTS_REQUIRE(p.ldx % 4 == 0 && p.ldo % 4 == 0, QTTS_ERR_ARG, "skinny: ldx/ldo % 4");
    QTTS_REQUIRE(!p.x_bf16 || (bf16 && p.ldx % 8 == 0), QTTS_ERR_ARG, "skinny: bf16 x needs the bf16 kernel and ldx % 8");
    QTTS_REQUIRE(!p.out_bf16 || bf16, QTTS_ERR_ARG, "skinny: bf16 output only in bf16 mode");
    QTTS_REQUIRE(bf16 || !p.norm || p.ss_in || skinny_f32_inline_norm(p.M, p.K), QTTS_ERR_ARG,
                 "skinny: the generic fp32 kernel takes the row sums of squares from ss_in");
    if (p.act == ACT_SWIGLU) QTTS_REQUIRE(p.N % 32 == 0 && fs == 16, QTTS_ERR_ARG, "skinny: swiglu needs N % 32 and fs == 16");
    if (p.act == ACT_SWIGLU8) {          // one 16-feature strip = 8 gate + 8 up rows: only the batch <= 8 kernel has this epilogue
        QTTS_REQUIRE(bf16 && fs == 16 && p.x_bf16 && p.M <= 8 && p.K % 512 == 0 && !p.bias && !QTTS_ABL(p, 15), QTTS_ERR_ARG,
                     "skinny: ACT_SWIGLU8 needs the bf16 batch <= 8 kernel (bf16 x, K % 512 == 0, 16-feature strips, no bias)");
        const char* e8 = QTTS_ENV("QTTS_SKINNY8");
        QTTS_REQUIRE(!(e8 && e8[0] == '0'), QTTS_ERR_ARG, "skinny: ACT_SWIGLU8 with QTTS_SKINNY8=0");
        if (launch8_fs<1, 16>(p, st)) { QTTS_CHECK_HIP(hipGetLastError()); return; }
        throw Error(QTTS_ERR_ARG, "skinny: no batch <= 8 instantiation for this K");
    }
    const int spw = skinny_spw(p.N, fs, p.act == ACT_SWIGLU);
    const int nw = (p.K / KT >= 16) ? 8 : 4;     // 8 waves split K unless K is tiny
    const int mt = p.M <= 16 ? 1 : (p.M <= 32 ? 2 : 4);
    if (bf16 && nw == 8 && mt == 1 && launch_skinny8(p, spw, fs, st)) { QTTS_CHECK_HIP(hipGetLastError()); return; }
    if (bf16 && mt == 2 && launch_skinny_ks(p, fs, st)) { QTTS_CHECK_HIP(hipGetLastError()); return; }
    if (!bf16 && !QTTS_ABL(p, 15) && launch_skinny8_f32(p, spw, st)) { QTTS_CHECK_HIP(hipGetLastError()); return; }
    if (bf16) {
        if (nw == 8) { if (mt == 1) launch2_mt<1, 8>(p, spw, fs, st); else if (mt == 2) launch2_mt<2, 8>(p, spw, fs, st); else launch2_mt<4, 8>(p, spw, fs, st); }
        else         { if (mt == 1) launch2_mt<1, 4>(p, spw, fs, st); else if (mt == 2) launch2_mt<2, 4>(p, spw, fs, st); else launch2_mt<4, 4>(p, spw, fs, st); }
    } else {
        QTTS_REQUIRE(!p.norm || p.ss_in, QTTS_ERR_ARG, "skinny: the generic fp32 kernel takes the row sums of squares from ss_in");
        if (mt == 1) launch_f32_mt<1>(p, spw, nw, st);
        else if (mt == 2) launch_f32_mt<2>(p, spw, nw, st);
        else launch_f32_mt<4>(p, spw, nw, st);
    }
    QTTS_CHECK_HIP(hipGetLastError());
}

size_t skinny_packed_bytes(int N, int K, bool bf16) { return (size_t)N * K * (bf16 ? 2 : 4); }

// Pack W[N][K] (row-major f32), optionally scaled per column by g[K] (folded RMSNorm weight), into the tile layout
// [N/fs strips][K/KT k-tiles][4 k-slices][fs features][16 B].
void pack_skinny_weight(const float* W, int N, int K, bool bf16, void* out_host, const float* g, int fs) {
    const int KT = bf16 ? 32 : 16;
    const int nkt = K / KT, strips = N / fs;
    parallel_for(strips, [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s)
            for (int kt = 0; kt < nkt; ++kt)
                for (int q = 0; q < 4; ++q)
                    for (int i = 0; i < fs; ++i) {
                        const size_t tile = ((size_t)s * nkt + kt) * (fs * 4) + q * fs + i;
                        const int k0 = kt * KT + q * (bf16 ? 8 : 4);
                        const float* src = W + (size_t)(s * fs + i) * K + k0;
                        if (bf16) {
                            bf16_t* d = reinterpret_cast<bf16_t*>(out_host) + tile * 8;
                            for (int e = 0; e < 8; ++e) d[e] = f32_to_bf16(g ? src[e] * g[k0 + e] : src[e]);
                        } else {
                            float* d = reinterpret_cast<float*>(out_host) + tile * 4;
                            for (int e = 0; e < 4; ++e) d[e] = g ? src[e] * g[k0 + e] : src[e];
                        }
                    }
    });
}

}  // namespace qtts
