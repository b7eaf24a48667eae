// sampling.hip -- on-device HF logits processors + sampler, and the frame-step glue kernels.
//
// Processor order and formulas are those of transformers 4.57.3 `_get_logits_processor`
// (SURVEY.md 3.3): RepetitionPenalty -> MinNewTokensLength -> SuppressTokens ->
// [Temperature -> TopK] -> argmax | softmax + multinomial.  argmax breaks ties towards the lowest
// index (torch.argmax).  Sampling draws from a counter-based Philox4x32-10 stream keyed by
// (seed; step, row, codebook) -- torch's global RNG stream is not reproducible across devices, so
// sampled runs are compared distributionally, greedy runs bit-exactly.
#include "common.h"
#include "kernels.h"
#include "tstamp.h"
#include "glue.h"

QTTS_TS_UNIT(sample)

namespace qtts {

__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                     uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ inline uint32_t float_key(float f) {  // monotone float -> uint (larger float = larger key)
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int SAMPLE_MAX_V = 8192;
constexpr int CAND_MAX = 1024;          // top-k candidates (scores >= the k-th largest per-thread max) ranked exactly in LDS

// The next pass's input rows (tabulated projected embedding, tabulated layer-0 q|k|v) are random rows of 117 / 470 MB tables:
// every request is an HBM round trip (~1 us), so ALL of a thread's requests are issued before the first store -- the round-1
// loops waited for each 16-byte piece in turn (up to 5 dependent round trips at the end of every code-predictor pass).
__device__ __forceinline__ void gather_next_rows(const SampleParams& p, int token, int b, int tid) {
    constexpr int G1 = 2, G2 = 4;                           // 16-byte pieces per thread: rows of <= 2048 / <= 4096 floats
    float4 v1[G1], v2[G2];
    const float* src = p.gather_emb ? p.gather_emb + (size_t)token * p.gather_C : nullptr;
    const float* src2 = p.gather2_emb ? p.gather2_emb + (size_t)token * p.gather2_C : nullptr;
#pragma unroll
    for (int it = 0; it < G1; ++it) {
        const int c = tid * 4 + it * 1024;
        v1[it] = (src && c < p.gather_C) ? *reinterpret_cast<const float4*>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < G2; ++it) {
        const int c = tid * 4 + it * 1024;
        v2[it] = (src2 && c < p.gather2_C) ? *reinterpret_cast<const float4*>(src2 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (src) {
#pragma unroll
        for (int it = 0; it < G1; ++it) {
            const int c = tid * 4 + it * 1024;
            if (c >= p.gather_C) continue;
            *reinterpret_cast<float4*>(p.gather_out + (size_t)b * p.gather_C + c) = v1[it];
            if (p.gather_out16) {
                ushort4 h; h.x = f32_to_bf16(v1[it].x); h.y = f32_to_bf16(v1[it].y); h.z = f32_to_bf16(v1[it].z); h.w = f32_to_bf16(v1[it].w);
                *reinterpret_cast<ushort4*>(p.gather_out16 + (size_t)b * p.gather_C + c) = h;
            }
        }
        for (int c = tid * 4 + G1 * 1024; c < p.gather_C; c += 1024) {          // (wider rows: the plain loop)
            const float4 v = *reinterpret_cast<const float4*>(src + c);
            *reinterpret_cast<float4*>(p.gather_out + (size_t)b * p.gather_C + c) = v;
            if (p.gather_out16) {
                ushort4 h; h.x = f32_to_bf16(v.x); h.y = f32_to_bf16(v.y); h.z = f32_to_bf16(v.z); h.w = f32_to_bf16(v.w);
                *reinterpret_cast<ushort4*>(p.gather_out16 + (size_t)b * p.gather_C + c) = h;
            }
        }
    }
    if (src2) {
#pragma unroll
        for (int it = 0; it < G2; ++it) {
            const int c = tid * 4 + it * 1024;
            if (c < p.gather2_C) *reinterpret_cast<float4*>(p.gather2_out + (size_t)b * p.gather2_C + c) = v2[it];
        }
        for (int c = tid * 4 + G2 * 1024; c < p.gather2_C; c += 1024)
            *reinterpret_cast<float4*>(p.gather2_out + (size_t)b * p.gather2_C + c) = *reinterpret_cast<const float4*>(src2 + c);
    }
}

__global__ __launch_bounds__(256) void sample_kernel(SampleParams p) {
    if (p.done_in && *p.done_in) return;
    __shared__ float sc[SAMPLE_MAX_V];
    __shared__ float fred[4];
    __shared__ int ired[8];
    __shared__ __attribute__((aligned(16))) float scan[256];
    __shared__ float cval[CAND_MAX];
    __shared__ int cidx[CAND_MAX];
    __shared__ __attribute__((aligned(16))) uint32_t ckey[CAND_MAX + 4];
    __shared__ int pick_lo, pick_hi;

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = p.V;
    const float* lg = p.logits + (size_t)b * p.ld;
    const int n_gen = p.n_generated_dev ? *p.n_generated_dev : 0;
    for (int v = tid; v < V; v += 256) sc[v] = lg[v];
    __syncthreads();
    if (p.generated && p.repetition_penalty != 1.0f) {
        for (int i = tid; i < n_gen; i += 256) {
            const int tok = p.generated[(size_t)b * p.gen_stride + i];
            const float s = lg[tok];
            sc[tok] = s < 0.f ? s * p.repetition_penalty : s / p.repetition_penalty;  // same value for duplicates
        }
        __syncthreads();
    }
    if (p.eos >= 0 && n_gen < p.min_new_tokens && tid == 0) sc[p.eos] = -INFINITY;
    __syncthreads();
    if (p.suppress_mask) {
        for (int v = tid; v < V; v += 256)
            if (p.suppress_mask[v]) sc[v] = -INFINITY;
        __syncthreads();
    }

    int token = 0;
    if (!p.do_sample) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int v = tid; v < V; v += 256) {
            const float s = sc[v];
            if (s > bv || (s == bv && v < bi)) { bv = s; bi = v; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { fred[wave] = bv; ired[wave] = bi; }
        __syncthreads();
        bv = fred[0]; bi = ired[0];
        for (int w = 1; w < 4; ++w)
            if (fred[w] > bv || (fred[w] == bv && ired[w] < bi)) { bv = fred[w]; bi = ired[w]; }
        token = bi;
    } else {
        if (p.temperature != 1.0f) {
            for (int v = tid; v < V; v += 256) sc[v] = sc[v] / p.temperature;
            __syncthreads();
        }
        const uint32_t step = p.step_dev ? (uint32_t)*p.step_dev : 0u;
        uint32_t rnd[4];
        const unsigned long long seed = p.seed_dev ? *p.seed_dev : p.seed;
        philox4x32_10(step, (uint32_t)b, p.stream_id, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
        const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
        bool sampled = false;
        bool need_search = p.top_k > 0 && p.top_k < V;
        if (need_search && p.top_k <= 256) {
            // ---- fast top-k: the k-th largest score is >= the k-th largest PER-THREAD maximum, so only scores at or
            // above that bound can be in the top-k.  Compact those few candidates and rank them exactly.
            float tm = -INFINITY;
            for (int v = tid; v < V; v += 256) tm = fmaxf(tm, sc[v]);
            uint32_t* keys = reinterpret_cast<uint32_t*>(scan);
            keys[tid] = float_key(tm);
            __syncthreads();
            {
                const uint32_t mk = keys[tid];
                int rank = 0;
                for (int j = 0; j < 256; j += 4) {                // all-pairs rank, 16-B LDS reads
                    const uint4 k4 = *reinterpret_cast<const uint4*>(&keys[j]);
                    rank += (k4.x > mk) || (k4.x == mk && j < tid);
                    rank += (k4.y > mk) || (k4.y == mk && j + 1 < tid);
                    rank += (k4.z > mk) || (k4.z == mk && j + 2 < tid);
                    rank += (k4.w > mk) || (k4.w == mk && j + 3 < tid);
                }
                if (rank == p.top_k - 1) pick_hi = (int)mk;     // lower bound T0 (as a key)
            }
            __syncthreads();
            const uint32_t T0 = (uint32_t)pick_hi;
            int mycnt = 0;
            for (int v = tid; v < V; v += 256) mycnt += __popcll(__ballot(float_key(sc[v]) >= T0));
            if (lane == 0) ired[wave] = mycnt;
            __syncthreads();
            const int n_c = ired[0] + ired[1] + ired[2] + ired[3];
            if (n_c <= CAND_MAX) {
                int base = 0;
                for (int w = 0; w < wave; ++w) base += ired[w];
                for (int v = tid; v < V; v += 256) {          // deterministic order: wave, slice, lane
                    const bool keep = float_key(sc[v]) >= T0;
                    const unsigned long long m = __ballot(keep);
                    if (keep) {
                        const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                        cval[slot] = sc[v];
                        cidx[slot] = v;
                        ckey[slot] = float_key(sc[v]);
                    }
                    base += __popcll(m);
                }
                for (int i = n_c + tid; i < ((n_c + 3) & ~3); i += 256) ckey[i] = 0u;   // pad to a multiple of 4 (key 0 < any real key)
                __syncthreads();
                for (int i = tid; i < n_c; i += 256) {        // exact rank among the candidates (value desc, slot asc)
                    const uint32_t mk = ckey[i];
                    int rank = 0;
                    for (int j = 0; j < n_c; j += 4) {
                        const uint4 k4 = *reinterpret_cast<const uint4*>(&ckey[j]);
                        rank += (k4.x > mk) || (k4.x == mk && j < i);
                        rank += (k4.y > mk) || (k4.y == mk && j + 1 < i);
                        rank += (k4.z > mk) || (k4.z == mk && j + 2 < i);
                        rank += (k4.w > mk) || (k4.w == mk && j + 3 < i);
                    }
                    if (rank == p.top_k - 1) pick_lo = (int)mk;  // key of the k-th largest score
                }
                __syncthreads();
                const uint32_t thr = (uint32_t)pick_lo;          // HF TopK keeps every score >= it (ties included)
                __syncthreads();
                if (wave == 0) {       // softmax over the survivors + inverse-CDF draw, one wave
                    float mx = -INFINITY;
                    for (int i = lane; i < n_c; i += 64) mx = fmaxf(mx, cval[i]);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                    float tot = 0.f;
                    for (int i = lane; i < n_c; i += 64) {
                        const float e = ckey[i] >= thr ? expf(cval[i] - mx) : 0.f;
                        cval[i] = e;
                        tot += e;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
                    if (p.top_p < 1.0f) {
                        // HF TopPLogitsWarper after TopK: ascending cumulative softmax <= 1 - top_p is removed, i.e. a token
                        // stays iff the probability mass of everything ranked strictly above it is < top_p (the top token
                        // always stays).  Ranks are exact (value desc, slot asc); all-pairs over the <= CAND_MAX candidates.
                        float keep_e[CAND_MAX / 64];
                        float tot2 = 0.f;
#pragma unroll
                        for (int q = 0; q < CAND_MAX / 64; ++q) {
                            const int i = q * 64 + lane;
                            keep_e[q] = 0.f;
                            if (i < n_c && cval[i] > 0.f) {
                                const uint32_t mk = ckey[i];
                                float above = 0.f;
                                for (int j = 0; j < n_c; ++j) {
                                    const uint32_t kj = ckey[j];
                                    if (kj > mk || (kj == mk && j < i)) above += cval[j];
                                }
                                if (above < p.top_p * tot) keep_e[q] = cval[i];
                            }
                            tot2 += keep_e[q];
                        }
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) tot2 += __shfl_xor(tot2, o);
#pragma unroll
                        for (int q = 0; q < CAND_MAX / 64; ++q) {
                            const int i = q * 64 + lane;
                            if (i < n_c) cval[i] = keep_e[q];
                        }
                        tot = tot2;
                    }
                    const float target = u * tot;
                    float run = 0.f;
                    int pick = -1, last = 0;
                    for (int i0 = 0; i0 < n_c && pick < 0; i0 += 64) {
                        const int i = i0 + lane;
                        const float e = i < n_c ? cval[i] : 0.f;
                        float inc = e;                                  // inclusive wave scan
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(inc, o); if (lane >= o) inc += t; }
                        const unsigned long long nz = __ballot(e > 0.f);
                        if (nz) last = i0 + 63 - __clzll((long long)nz);
                        const unsigned long long hit = __ballot(e > 0.f && run + inc > target);
                        if (hit) pick = i0 + __ffsll((long long)hit) - 1;
                        run += __shfl(inc, 63);
                    }
                    if (pick < 0) pick = last;                          // rounding at the very top of the CDF
                    if (lane == 0) pick_lo = cidx[pick];
                }
                __syncthreads();
                token = pick_lo;
                sampled = true;
                need_search = false;
            }
        }
        if (need_search) {
            // ---- general top-k: k-th largest key by a bitwise binary search (atomics-free: ballot + popcount) ----
            // invariant: count(key >= prefix) >= k; set bits from the MSB down while that still holds.
            uint32_t prefix = 0;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = prefix | (1u << bit);
                int cnt = 0;
                for (int v = tid; v < V; v += 256) cnt += __popcll(__ballot(float_key(sc[v]) >= cand));
                if (lane == 0) ired[(bit & 1) * 4 + wave] = cnt;
                __syncthreads();
                const int* r4 = ired + (bit & 1) * 4;
                if (r4[0] + r4[1] + r4[2] + r4[3] >= p.top_k) prefix = cand;
            }
            __syncthreads();
            for (int v = tid; v < V; v += 256)
                if (float_key(sc[v]) < prefix) sc[v] = -INFINITY;
            __syncthreads();
        }
        if (!sampled) {
            // ---- general path: softmax numerators over the whole vocabulary, block scan, inverse CDF ----
            float m = -INFINITY;
            for (int v = tid; v < V; v += 256) m = fmaxf(m, sc[v]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) fred[wave] = m;
            __syncthreads();
            m = fmaxf(fmaxf(fred[0], fred[1]), fmaxf(fred[2], fred[3]));
            __syncthreads();
            const int chunk = (V + 255) / 256;
            const int v0 = tid * chunk, v1 = min(V, v0 + chunk);
            float mine = 0.f;
            for (int v = v0; v < v1; ++v) {
                const float e = expf(sc[v] - m);
                sc[v] = e;
                mine += e;
            }
            if (p.top_p < 1.0f) {
                // ---- general top-p (HF TopPLogitsWarper, IM:287-352 forwards any top_p): with no top-k bound, or one beyond the
                // candidate buffers above, the nucleus is cut on the whole vocabulary.  A token stays iff the softmax mass of everything
                // ranked strictly above it is < top_p (ascending cumulative probability > 1 - top_p; the top token always stays), so
                // the kept set is {e >= e*}: e* comes from a bit-by-bit search over the 32-bit key space for the LARGEST key c whose
                // strictly-above mass S(c) is still >= top_p * total -- 32 block-wide masked sums, one barrier each (exp is
                // monotone: keys of the numerators order like keys of the scores).  Equal scores are kept or cut together.
                float* pred = scan;                          // [2][4] partial sums, double-buffered by round parity
                auto block_sum = [&](float v, int slot) -> float {
                    v = wave_sum64_dpp(v);
                    if (lane == 0) pred[slot * 4 + wave] = v;
                    __syncthreads();
                    return (pred[slot * 4 + 0] + pred[slot * 4 + 1]) + (pred[slot * 4 + 2] + pred[slot * 4 + 3]);
                };
                const float total0 = block_sum(mine, 0);
                const float P = p.top_p * total0;
                uint32_t cut = 0u;                           // largest key with S(key) >= P found so far
                bool any_cut = false;
                for (int bit = 31; bit >= 0; --bit) {
                    const uint32_t cand = cut | (1u << bit);
                    float above = 0.f;
                    for (int v = v0; v < v1; ++v) above += float_key(sc[v]) > cand ? sc[v] : 0.f;
                    const float S = block_sum(above, 1 + (bit & 1));
                    if (S >= P) { cut = cand; any_cut = true; }
                }
                // (no candidate passed: only key 0 could still be cut, and a numerator e >= 0 has a key >= 2^31)
                __syncthreads();                             // `pred` aliases `scan`, which is rewritten below
                mine = 0.f;
                for (int v = v0; v < v1; ++v) {
                    if (any_cut && float_key(sc[v]) <= cut) sc[v] = 0.f;
                    mine += sc[v];
                }
            }
            scan[tid] = mine;
            if (tid == 0) { pick_lo = 0x7fffffff; pick_hi = -1; }
            __syncthreads();
            for (int o = 1; o < 256; o <<= 1) {  // inclusive Hillis-Steele scan in index order
                const float add = tid >= o ? scan[tid - o] : 0.f;
                __syncthreads();
                scan[tid] += add;
                __syncthreads();
            }
            const float total = scan[255];
            const float target = u * total;
            float run = scan[tid] - mine;
            for (int v = v0; v < v1; ++v) {
                const float e = sc[v];
                if (e > 0.f) {
                    atomicMax(&pick_hi, v);
                    if (run + e > target) { atomicMin(&pick_lo, v); break; }
                }
                run += e;
            }
            __syncthreads();
            token = pick_lo != 0x7fffffff ? pick_lo : pick_hi;
        }
    }

    if (token < 0 || token >= V) token = 0;   // all-NaN logits must not turn into an out-of-range gather index
    gather_next_rows(p, token, b, tid);        // every thread holds the same token (block-uniform by construction)
    if (tid == 0) {
        if (p.unfinished) {
            const int uf = p.unfinished[b];
            if (!uf) token = p.eos;                        // finished rows keep receiving pad (= eos)
            p.unfinished[b] = uf && (token != p.eos);
            if (p.generated_out) p.generated_out[(size_t)b * p.gen_stride + n_gen] = token;
        }
        p.tok_out[(size_t)b * p.tok_stride] = token;
    }
}

// Wave-level tail of the top-k sampler (round 2, second pass -- the in-kernel timestamps of profiles/r02_tstamp_frame.md showed
// 6.8 of the sampler's 12.1 us in the all-pairs ranking): this lane holds the candidates of slots lane, lane + 64, ... in
// registers (key 0 = no candidate: below every real key).  The key of the k-th largest score (HF TopK keeps every score >= it,
// ties included) comes from a bit-by-bit selection -- the largest v with #{key >= v} >= k -- one compare and one scalar
// popcount per slot and bit, no LDS traffic and no barrier; then the softmax numerators of the survivors go back to `cval`
// (slot order) and their sum is returned, accumulated in the order sample_kernel uses (slots ascending per lane, butterfly).
template <int NQ>
__device__ inline float topk_softmax_wave(int n_c, int top_k, int lane, float* cval, const uint32_t* ckey) {
    uint32_t kq[NQ];
    float vq[NQ];
    const int nq = (n_c + 63) >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = q * 64 + lane;
        kq[q] = i < n_c ? ckey[i] : 0u;
        vq[q] = i < n_c ? cval[i] : -INFINITY;
    }
    // (two bits per round: the three candidates' counts are independent, so a round costs one vector-compare -> scalar-popcount
    // round trip instead of two)
    uint32_t thr = 0u;
#pragma unroll 1
    for (int lo = 30; lo >= 0; lo -= 2) {
        const uint32_t c1 = thr | (1u << lo), c2 = thr | (2u << lo), c3 = thr | (3u << lo);
        int n1 = 0, n2 = 0, n3 = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (NQ <= 2 || q < nq) {                                              // (nq: wave-uniform, a scalar branch)
                n1 += __popcll(__ballot(kq[q] >= c1));
                n2 += __popcll(__ballot(kq[q] >= c2));
                n3 += __popcll(__ballot(kq[q] >= c3));
            }
        thr = n3 >= top_k ? c3 : (n2 >= top_k ? c2 : (n1 >= top_k ? c1 : thr));
    }
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < NQ; ++q) mx = fmaxf(mx, vq[q]);
    mx = wave_max64_dpp(mx);
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i = q * 64 + lane;
        if (i < n_c) {
            const float e = kq[q] >= thr ? expf(vq[q] - mx) : 0.f;
            cval[i] = e;
            tot += e;
        }
    }
    return wave_sum64_dpp(tot);           // (rows summed on DPP rotations, the four row sums in a fixed order)
}

// Round 2 (measured 1.1 % faster per frame, profiles/r02_ab_variants.md, now the default): the sampling path for 0 < top_k <= 64 and V <= 4096 (the
// reference default is top_k = 50 on 2048 / 3072 logits) with the fixed costs taken out of sample_kernel above:
//   * every load that does not depend on another load (done flag, counters, Philox key, the row's logits, the suppress
//     mask) is issued at kernel entry instead of behind the early-exit test and the processors' barriers;
//   * the processors run on the thread's own V / 256 logits in registers (one barrier instead of four; the repetition
//     penalty, which is a scatter, goes through presence flags in LDS -- talker call only);
//   * the candidate bound is the k-th largest of 64 quad maxima (each wave ranks the 64 keys itself: 16 LDS reads and no
//     second barrier) instead of an all-pairs rank over 256 thread maxima.  The bound is looser (about 1.7x the
//     candidates) but the exact ranking below it is unchanged.
// The surviving candidates keep the slot order of sample_kernel (wave, slice, lane), so for the same Philox draw the same
// token comes out (up to rounding in the inverse-CDF scan); the greedy / parity path never comes here.
template <int EPT>
__global__ __launch_bounds__(256) void sample_kernel_v2(SampleParams p) {
    __shared__ float sc[EPT * 256];
    __shared__ float fred[4];
    __shared__ int ired[8];
    __shared__ __attribute__((aligned(16))) float scan[256];
    __shared__ float cval[CAND_MAX];
    __shared__ int cidx[CAND_MAX];
    __shared__ __attribute__((aligned(16))) uint32_t ckey[CAND_MAX + 4];
    __shared__ __attribute__((aligned(16))) uint32_t gkey[64];
    __shared__ int pick_lo, pick_hi;

    QTTS_TS_BEGIN();
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int V = p.V;
    const float* lg = p.logits + (size_t)b * p.ld;
    // ---- 0. independent loads, all in flight together
    const int done = p.done_in ? *p.done_in : 0;
    const int n_gen = p.n_generated_dev ? *p.n_generated_dev : 0;
    const uint32_t step = p.step_dev ? (uint32_t)*p.step_dev : 0u;
    const unsigned long long seed = p.seed_dev ? *p.seed_dev : p.seed;
    // the first 256 entries of this row's token history (repetition penalty; the talker's call only) are requested here, before the
    // count is known: behind the flag-clearing barrier below the load was a memory round trip of its own (2-6 us by box)
    const int* gsrc = p.generated ? p.generated + (size_t)b * p.gen_stride + (tid < p.gen_stride ? tid : 0) : reinterpret_cast<const int*>(lg);
    const int gtok0 = *gsrc;
    float x[EPT];
    unsigned char sup[EPT];
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
        const int v = it * 256 + tid;
        x[it] = v < V ? lg[v] : -INFINITY;
        sup[it] = (p.suppress_mask && v < V) ? p.suppress_mask[v] : (unsigned char)0;
    }
    if (done) return;
    QTTS_TS_DRAINED(1);                    // (tstamp build: phases 1..5 = logits arrived | bound | candidates ranked | drawn | rows gathered)
    // ---- 1. HF processors, in HF's order, on this thread's own logits
    if (p.generated && p.repetition_penalty != 1.0f) {       // scatter -> presence flags in LDS (idempotent for duplicates)
#pragma unroll
        for (int it = 0; it < EPT; ++it) sc[it * 256 + tid] = 0.f;
        __syncthreads();
        if (tid < n_gen && gtok0 >= 0 && gtok0 < V) sc[gtok0] = 1.f;          // entries 0..255: prefetched at entry
        for (int i = tid + 256; i < n_gen; i += 256) {
            const int tok = p.generated[(size_t)b * p.gen_stride + i];
            if (tok >= 0 && tok < V) sc[tok] = 1.f;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < EPT; ++it)
            if (sc[it * 256 + tid] != 0.f) x[it] = x[it] < 0.f ? x[it] * p.repetition_penalty : x[it] / p.repetition_penalty;
        __syncthreads();                                     // flags consumed before the scores overwrite them
    }
    const bool block_eos = p.eos >= 0 && n_gen < p.min_new_tokens;
    float tm = -INFINITY;
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
        const int v = it * 256 + tid;
        if ((block_eos && v == p.eos) || sup[it]) x[it] = -INFINITY;
        if (p.temperature != 1.0f) x[it] = x[it] / p.temperature;
        sc[v] = x[it];                                       // for the general fallback paths only
        if (v < V) tm = fmaxf(tm, x[it]);
    }
    uint32_t rnd[4];
    philox4x32_10(step, (uint32_t)b, p.stream_id, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), rnd);
    const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f);
    // ---- 2. candidate bound: k-th largest of the 64 quad maxima (<= the k-th largest score), ranked by every wave itself
    tm = fmaxf(tm, __shfl_xor(tm, 1));
    tm = fmaxf(tm, __shfl_xor(tm, 2));
    if ((lane & 3) == 0) gkey[wave * 16 + (lane >> 2)] = float_key(tm);
    __syncthreads();
    QTTS_TS(2);
    // (bit-by-bit selection on the wave's 64 keys: the largest value v with #{key >= v} >= k IS the k-th largest key; one
    // compare + one scalar popcount per bit, no LDS traffic and a twentieth of the all-pairs rank's instructions)
    uint32_t T0 = 0u;
    {
        const uint32_t mk = gkey[lane];
#pragma unroll 1
        for (int lo = 30; lo >= 0; lo -= 2) {                  // two bits per round (see topk_softmax_wave)
            const uint32_t c1 = T0 | (1u << lo), c2 = T0 | (2u << lo), c3 = T0 | (3u << lo);
            const int n1 = __popcll(__ballot(mk >= c1)), n2 = __popcll(__ballot(mk >= c2)), n3 = __popcll(__ballot(mk >= c3));
            T0 = n3 >= p.top_k ? c3 : (n2 >= p.top_k ? c2 : (n1 >= p.top_k ? c1 : T0));
        }
    }
    // ---- 3. compaction in sample_kernel's slot order (wave, slice, lane)
    unsigned long long mb[EPT];
    int mycnt = 0;
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
        const int v = it * 256 + tid;
        mb[it] = __ballot(v < V && float_key(x[it]) >= T0);
        mycnt += __popcll(mb[it]);
    }
    if (lane == 0) ired[wave] = mycnt;
    __syncthreads();
    const int n_c = ired[0] + ired[1] + ired[2] + ired[3];
    int token = 0;
    bool sampled = false;
    if (n_c <= CAND_MAX) {
        int base = 0;
        for (int w = 0; w < wave; ++w) base += ired[w];
#pragma unroll
        for (int it = 0; it < EPT; ++it) {
            if ((mb[it] >> lane) & 1ull) {
                const int slot = base + __popcll(mb[it] & ((1ull << lane) - 1ull));
                cval[slot] = x[it];
                cidx[slot] = it * 256 + tid;
                ckey[slot] = float_key(x[it]);
            }
            base += __popcll(mb[it]);
        }
        __syncthreads();
        QTTS_TS(3);
        if (wave == 0) {       // exact top-k threshold, softmax over the survivors and the inverse-CDF draw: one wave, no barrier
            float tot = n_c <= 128 ? topk_softmax_wave<2>(n_c, p.top_k, lane, cval, ckey)      // (typical: ~1.7 k candidates)
                                   : topk_softmax_wave<CAND_MAX / 64>(n_c, p.top_k, lane, cval, ckey);
            if (p.top_p < 1.0f) {
                float keep_e[CAND_MAX / 64];
                float tot2 = 0.f;
#pragma unroll
                for (int q = 0; q < CAND_MAX / 64; ++q) {
                    const int i = q * 64 + lane;
                    keep_e[q] = 0.f;
                    if (i < n_c && cval[i] > 0.f) {
                        const uint32_t mk = ckey[i];
                        float above = 0.f;
                        for (int j = 0; j < n_c; ++j) {
                            const uint32_t kj = ckey[j];
                            if (kj > mk || (kj == mk && j < i)) above += cval[j];
                        }
                        if (above < p.top_p * tot) keep_e[q] = cval[i];
                    }
                    tot2 += keep_e[q];
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) tot2 += __shfl_xor(tot2, o);
#pragma unroll
                for (int q = 0; q < CAND_MAX / 64; ++q) {
                    const int i = q * 64 + lane;
                    if (i < n_c) cval[i] = keep_e[q];
                }
                tot = tot2;
            }
            const float target = u * tot;
            float run = 0.f;
            int pick = -1, last = 0;
            for (int i0 = 0; i0 < n_c && pick < 0; i0 += 64) {
                const int i = i0 + lane;
                const float e = i < n_c ? cval[i] : 0.f;
                const float inc = wave_incl_scan64_dpp(e, lane);
                const unsigned long long nz = __ballot(e > 0.f);
                if (nz) last = i0 + 63 - __clzll((long long)nz);
                const unsigned long long hit = __ballot(e > 0.f && run + inc > target);
                if (hit) pick = i0 + __ffsll((long long)hit) - 1;
                run += lane_bcast(inc, 63);
            }
            if (pick < 0) pick = last;
            if (lane == 0) pick_lo = cidx[pick];
        }
        __syncthreads();
        QTTS_TS(4);
        token = pick_lo;
        sampled = true;
    }
    if (!sampled) {
        // ---- more than CAND_MAX scores at or above the bound (massive ties): the general paths of sample_kernel on `sc`
        uint32_t prefix = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = prefix | (1u << bit);
            int cnt = 0;
            for (int v = tid; v < V; v += 256) cnt += __popcll(__ballot(float_key(sc[v]) >= cand));
            if (lane == 0) ired[(bit & 1) * 4 + wave] = cnt;
            __syncthreads();
            const int* r4 = ired + (bit & 1) * 4;
            if (r4[0] + r4[1] + r4[2] + r4[3] >= p.top_k) prefix = cand;
        }
        __syncthreads();
        for (int v = tid; v < V; v += 256)
            if (float_key(sc[v]) < prefix) sc[v] = -INFINITY;
        __syncthreads();
        float m = -INFINITY;
        for (int v = tid; v < V; v += 256) m = fmaxf(m, sc[v]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) fred[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(fred[0], fred[1]), fmaxf(fred[2], fred[3]));
        __syncthreads();
        const int chunk = (V + 255) / 256;
        const int v0 = tid * chunk, v1 = min(V, v0 + chunk);
        float mine = 0.f;
        for (int v = v0; v < v1; ++v) {
            const float e = expf(sc[v] - m);
            sc[v] = e;
            mine += e;
        }
        scan[tid] = mine;
        if (tid == 0) { pick_lo = 0x7fffffff; pick_hi = -1; }
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const float add = tid >= o ? scan[tid - o] : 0.f;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        const float total = scan[255];
        const float target = u * total;
        float run = scan[tid] - mine;
        for (int v = v0; v < v1; ++v) {
            const float e = sc[v];
            if (e > 0.f) {
                atomicMax(&pick_hi, v);
                if (run + e > target) { atomicMin(&pick_lo, v); break; }
            }
            run += e;
        }
        __syncthreads();
        token = pick_lo != 0x7fffffff ? pick_lo : pick_hi;
    }

    if (token < 0 || token >= V) token = 0;
    gather_next_rows(p, token, b, tid);
    if (tid == 0) {
        if (p.unfinished) {
            const int uf = p.unfinished[b];
            if (!uf) token = p.eos;
            p.unfinished[b] = uf && (token != p.eos);
            if (p.generated_out) p.generated_out[(size_t)b * p.gen_stride + n_gen] = token;
        }
        p.tok_out[(size_t)b * p.tok_stride] = token;
    }
    QTTS_TS_DRAINED(5);
    QTTS_TS_END(sample, 3, V, 0);
}

// (Round 3 measured a ONE-WAVE-per-row sampler for the code predictor's call shape -- 32 logits per lane in registers, no workgroup
// barrier, candidates compacted by 32 ballots: 14.2 us per launch against 8.5 us for sample_kernel_v2, frame 2.76 vs 2.68 ms
// (profiles/r03_ab_sampler_w1.md).  One wave issuing every load, ballot and row gather of the launch is slower than four waves with
// three barriers; deleted.)
void launch_sample(const SampleParams& p, hipStream_t st) {
    QTTS_REQUIRE(p.V <= SAMPLE_MAX_V, QTTS_ERR_LIMIT, "sample: vocab too large");
    QTTS_REQUIRE(!(p.do_sample && p.top_p < 1.0f) || p.top_p > 0.0f, QTTS_ERR_ARG, "sample: top_p must be in (0, 1]");
    if (p.do_sample && p.top_k > 0 && p.top_k <= 64 && p.top_k < p.V && p.V <= 4096) {
        if (p.V <= 2048) hipLaunchKernelGGL(sample_kernel_v2<8>, dim3(p.B), dim3(256), 0, st, p);
        else if (p.V <= 3072) hipLaunchKernelGGL(sample_kernel_v2<12>, dim3(p.B), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(sample_kernel_v2<16>, dim3(p.B), dim3(256), 0, st, p);
        QTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(sample_kernel, dim3(p.B), dim3(256), 0, st, p);
    QTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------- loop bookkeeping
// After every row sampled token #n: n_generated++, kv_len++, and latch `done` exactly where HF's
// stopping criteria would break (all rows finished, or max_new_tokens reached).
__global__ void sample_finish_kernel(StepState st, int B, int max_new_tokens) {
    if (*st.done) return;
    if (threadIdx.x == 0) {
        const int n = *st.n_generated + 1;
        *st.n_generated = n;
        *st.gen_step += 1;
        *st.kv_len += 1;
        if (st.frame_serial) *st.frame_serial += 1;
        int any = 0;
        for (int b = 0; b < B; ++b) any |= st.unfinished[b];
        if (n >= max_new_tokens || !any) { *st.done = 1; *st.final_count = n; }
    }
}
void launch_sample_finish(const StepState& s, int B, int max_new_tokens, hipStream_t st) {
    hipLaunchKernelGGL(sample_finish_kernel, dim3(1), dim3(64), 0, st, s, B, max_new_tokens);
    QTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------- teacher forcing (diagnostic mode)
// Frame-level teacher forcing for the parity measurements of the bf16 mode (tests/test_gpu_parity.py): the engine's own greedy
// choices are RECORDED and then REPLACED by a given code sequence, so that a whole utterance is compared decision by decision
// instead of stopping at the first flip.  Three tiny kernels, launched only in this mode (never in the captured frame graph).
__global__ void teacher_trace_kernel(TeacherParams p) {          // before a cb-0 sample: raw logits of selected token steps
    if (*p.st.done) return;
    const int i = *p.st.n_generated;                                 // index of the token about to be sampled
    const int slot = (p.slots && i <= p.F) ? p.slots[i] : -1;
    if (slot < 0) return;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < p.V; c += blockDim.x) p.trace[((size_t)slot * p.B + b) * p.V + c] = p.logits[(size_t)b * p.V + c];
}
__global__ void teacher_tok_kernel(TeacherParams p) {            // after a cb-0 sample + sample_finish (idempotent once `done`)
    const int b = threadIdx.x;
    if (b >= p.B) return;
    const int i = *p.st.n_generated - 1;                             // index of the token just sampled
    if (i < 0 || i > p.F) return;
    p.own[((size_t)b * (p.F + 1) + i) * p.G] = p.cur_tok[b];
    if (i < p.F) {
        const int tok = (int)p.codes[((size_t)b * p.F + i) * p.G];
        p.cur_tok[b] = tok;
        p.generated[(size_t)b * p.gen_stride + i] = tok;            // the repetition-penalty history follows the forced sequence
    }
}
__global__ void teacher_sub_kernel(TeacherParams p) {            // after the code predictor's passes, before the embedding sum
    if (*p.st.done) return;
    const int b = blockIdx.x, j = threadIdx.x;
    const int f = *p.st.gen_step;
    if (j >= p.G - 1 || f >= p.F) return;
    p.own[((size_t)b * (p.F + 1) + f) * p.G + 1 + j] = p.sub[(size_t)b * p.sub_stride + j];
    p.sub[(size_t)b * p.sub_stride + j] = (int)p.codes[((size_t)b * p.F + f) * p.G + 1 + j];
}
void launch_teacher(const TeacherParams& p, int which, hipStream_t st) {
    QTTS_REQUIRE(p.B <= 64 && p.G <= 64, QTTS_ERR_LIMIT, "teacher forcing: B, G <= 64");
    if (which == 0) hipLaunchKernelGGL(teacher_trace_kernel, dim3(p.B), dim3(256), 0, st, p);
    else if (which == 1) hipLaunchKernelGGL(teacher_tok_kernel, dim3(1), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(teacher_sub_kernel, dim3(p.B), dim3(64), 0, st, p);
    QTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------- glue kernels
__device__ inline float block_sum256(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// code-predictor input rows (M:1671-1672, 1281).  pass 0: rows [0,B) = past_hidden, rows [B,2B) =
// talker codec_embedding[cur_tok]; pass j>0: rows [0,B) = cp codec_embedding[j-1][sub[b][j-1]].
__global__ __launch_bounds__(256) void cp_gather_kernel(CpGatherParams p) {
    if (p.done && *p.done) return;
    const int r = blockIdx.x;
    const float* src;
    if (p.pass == 0) {
        const int t = r / p.B, b = r % p.B;
        src = t == 0 ? p.past_hidden + (size_t)b * p.H : p.talker_emb + (size_t)p.cur_tok[b] * p.H;
    } else {
        src = p.cp_emb + ((size_t)(p.pass - 1) * p.cp_vocab + p.sub[(size_t)r * p.sub_stride + p.pass - 1]) * p.H;
    }
    for (int c = threadIdx.x * 4; c < p.H; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(src + c);
        *reinterpret_cast<float4*>(p.out + (size_t)r * p.H + c) = v;
        if (p.out16) {
            ushort4 h; h.x = f32_to_bf16(v.x); h.y = f32_to_bf16(v.y); h.z = f32_to_bf16(v.z); h.w = f32_to_bf16(v.w);
            *reinterpret_cast<ushort4*>(p.out16 + (size_t)r * p.H + c) = h;
        }
    }
}
void launch_cp_gather(const CpGatherParams& p, hipStream_t st) {
    hipLaunchKernelGGL(cp_gather_kernel, dim3(p.pass == 0 ? 2 * p.B : p.B), dim3(256), 0, st, p);
    QTTS_CHECK_HIP(hipGetLastError());
}

// next talker input (M:1681-1692): sum of the 16 codebook embeddings (+ trailing text or tts_pad), and
// the frame's outputs: codes[b][f][:] (int64) and hidden[b][f] = past_hidden.
__global__ __launch_bounds__(256) void embed_sum_kernel(EmbedSumParams p) {
    // (round 2: all code indices first, then all 16 embedding rows of a column block in flight together -- unconditional requests
    // with clamped indices, summed in codebook order as before.  The loop `tk = sub[i]; a += emb[i][tk]` was 15 dependent pairs of
    // memory round trips: 16.7 us per frame in the kernel trace)
    constexpr int GMAX = 16;
    const int b = blockIdx.x;
    const int ncp = p.G - 1, nfast = ncp < GMAX ? ncp : GMAX;
    int tk[GMAX];
#pragma unroll
    for (int i = 0; i < GMAX; ++i) tk[i] = p.sub[(size_t)b * p.sub_stride + (i < ncp ? i : 0)];
    const int done = *p.st.done;
    const int f = *p.st.gen_step;          // frame index == generation_step
    const int tok0 = p.cur_tok[b];
    if (done) return;
    for (int c = threadIdx.x * 4; c < p.H; c += 1024) {
        float4 e[GMAX];
#pragma unroll
        for (int i = 0; i < GMAX; ++i)
            e[i] = *reinterpret_cast<const float4*>(p.cp_emb + ((size_t)(i < ncp ? i : 0) * p.cp_vocab + tk[i]) * p.H + c);
        float4 a = *reinterpret_cast<const float4*>(p.talker_emb + (size_t)tok0 * p.H + c);
        const float* tp = f < p.Tt ? p.trailing + ((size_t)b * p.Tt + f) * p.H : p.tts_pad;
        const float4 t = *reinterpret_cast<const float4*>(tp + c);
#pragma unroll
        for (int i = 0; i < GMAX; ++i)
            if (i < nfast) { a.x += e[i].x; a.y += e[i].y; a.z += e[i].z; a.w += e[i].w; }
        for (int i = GMAX; i < ncp; ++i) {   // (more than 17 code groups: the remaining tables one by one)
            const int tki = p.sub[(size_t)b * p.sub_stride + i];
            const float4 ei = *reinterpret_cast<const float4*>(p.cp_emb + ((size_t)i * p.cp_vocab + tki) * p.H + c);
            a.x += ei.x; a.y += ei.y; a.z += ei.z; a.w += ei.w;
        }
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        *reinterpret_cast<float4*>(p.x_out + (size_t)b * p.H + c) = a;
        if (p.x_out16) {
            ushort4 h; h.x = f32_to_bf16(a.x); h.y = f32_to_bf16(a.y); h.z = f32_to_bf16(a.z); h.w = f32_to_bf16(a.w);
            *reinterpret_cast<ushort4*>(p.x_out16 + (size_t)b * p.H + c) = h;
        }
        if (p.hidden_out) {
            const float4 h = *reinterpret_cast<const float4*>(p.past_hidden + (size_t)b * p.H + c);
            *reinterpret_cast<float4*>(p.hidden_out + ((size_t)b * p.max_frames + f) * p.H + c) = h;
        }
    }
    if (threadIdx.x < p.G && f < p.max_frames) {
        const int64_t v = threadIdx.x == 0 ? tok0 : p.sub[(size_t)b * p.sub_stride + threadIdx.x - 1];
        p.codes_out[((size_t)b * p.max_frames + f) * p.G + threadIdx.x] = v;
    }
}
void launch_embed_sum(const EmbedSumParams& p, hipStream_t st) {
    hipLaunchKernelGGL(embed_sum_kernel, dim3(p.B), dim3(256), 0, st, p);
    QTTS_CHECK_HIP(hipGetLastError());
}

// y = g * (x * rsqrt(mean(x^2) + eps)) (Qwen3TTSRMSNorm M:605-610) for the final talker norm -> past_hidden
__global__ __launch_bounds__(256) void apply_norm_kernel(const float* x, int ldx, const float* g, float eps, float* y,
                                                         int ldy, int C, const int* done, unsigned short* y16) {
    if (done && *done) return;
    __shared__ float sm[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = block_sum256(s, sm);
    const float rstd = rsqrtf(s / (float)C + eps);
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
        const float4 w = *reinterpret_cast<const float4*>(g + c);
        float4 o;
        o.x = w.x * (v.x * rstd); o.y = w.y * (v.y * rstd); o.z = w.z * (v.z * rstd); o.w = w.w * (v.w * rstd);
        *reinterpret_cast<float4*>(y + (size_t)r * ldy + c) = o;
        if (y16) {                         // bf16 copy for the consuming GEMM (codec_head)
            ushort4 h; h.x = f32_to_bf16(o.x); h.y = f32_to_bf16(o.y); h.z = f32_to_bf16(o.z); h.w = f32_to_bf16(o.w);
            *reinterpret_cast<ushort4*>(y16 + (size_t)r * ldy + c) = h;
        }
    }
}
void launch_apply_norm(const float* x, int ldx, const float* g, float eps, float* y, int ldy, int rows, int C,
                       const int* done, hipStream_t st, unsigned short* y16) {
    hipLaunchKernelGGL(apply_norm_kernel, dim3(rows), dim3(256), 0, st, x, ldx, g, eps, y, ldy, C, done, y16);
    QTTS_CHECK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void row_ss_kernel(const float* x, int ldx, int C, float* ss, const int* done) {
    if (done && *done) return;
    __shared__ float sm[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = block_sum256(s, sm);
    if (threadIdx.x == 0) ss[r] = s;
}
void launch_row_ss(const float* x, int ldx, int rows, int C, float* ss, const int* done, hipStream_t st) {
    hipLaunchKernelGGL(row_ss_kernel, dim3(rows), dim3(256), 0, st, x, ldx, C, ss, done);
    QTTS_CHECK_HIP(hipGetLastError());
}


// ---- prompt assembly -----------------------------------------------------------------------------------------------
// One workgroup per output row.  The codec side of an ICL row is the 16-way codebook embedding sum in codebook order
// (M:1983-1990), then text + codec (commutative for two terms), exactly the reference's association.
__global__ __launch_bounds__(256) void assemble_rows_kernel(AssembleParams p) {
    const int r = blockIdx.x;
    const int tr = p.desc[r * 4 + 0], cid = p.desc[r * 4 + 1], sr = p.desc[r * 4 + 2], rf = p.desc[r * 4 + 3];
    bool bad = tr >= p.proj_rows || cid >= p.vocab || sr >= p.n_spk || rf >= p.n_ref;
    int64_t rc0 = 0;
    if (rf >= 0 && !bad) {
        rc0 = p.ref_codes[(size_t)rf * p.G];
        bad = rc0 < 0 || rc0 >= p.vocab;
        for (int g = 1; g < p.G; ++g) {
            const int64_t v = p.ref_codes[(size_t)rf * p.G + g];
            bad = bad || v < 0 || v >= p.cp_vocab;
        }
    }
    if (bad && threadIdx.x == 0) *p.err = 1;
    for (int c = threadIdx.x * 4; c < p.H; c += 1024) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        bool have = false;
        if (!bad) {
            if (cid >= 0) { a = *reinterpret_cast<const float4*>(p.talker_emb + (size_t)cid * p.H + c); have = true; }
            if (sr >= 0) {
                const float4 e = *reinterpret_cast<const float4*>(p.spk + (size_t)sr * p.H + c);
                if (have) { a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w; } else { a = e; have = true; }
            }
            if (rf >= 0) {
                float4 e = *reinterpret_cast<const float4*>(p.talker_emb + (size_t)rc0 * p.H + c);
                for (int g = 1; g < p.G; ++g) {
                    const int64_t v = p.ref_codes[(size_t)rf * p.G + g];
                    const float4 f = *reinterpret_cast<const float4*>(p.cp_emb + ((size_t)(g - 1) * p.cp_vocab + v) * p.H + c);
                    e.x += f.x; e.y += f.y; e.z += f.z; e.w += f.w;
                }
                if (have) { a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w; } else { a = e; have = true; }
            }
            if (tr >= 0) {
                const float4 t = *reinterpret_cast<const float4*>(p.proj + (size_t)tr * p.H + c);
                if (have) { a.x = t.x + a.x; a.y = t.y + a.y; a.z = t.z + a.z; a.w = t.w + a.w; } else a = t;
            }
        }
        *reinterpret_cast<float4*>(p.out + (size_t)r * p.H + c) = a;
    }
}
void launch_assemble_rows(const AssembleParams& p, hipStream_t st) {
    QTTS_REQUIRE(p.H % 4 == 0, QTTS_ERR_ARG, "assemble: H % 4");
    hipLaunchKernelGGL(assemble_rows_kernel, dim3(p.rows), dim3(256), 0, st, p);
    QTTS_CHECK_HIP(hipGetLastError());
}

template <bool BF16>
__global__ __launch_bounds__(256) void gather_rows_kernel(const void* table, int64_t n_table, int C, const int64_t* ids, float* out,
                                                          int* err) {
    const int r = blockIdx.x;
    const int64_t id = ids[r];
    const bool bad = id < 0 || id >= n_table;
    if (bad && threadIdx.x == 0) *err = 1;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!bad) {
            if constexpr (BF16) {
                const ushort4 h = *reinterpret_cast<const ushort4*>(reinterpret_cast<const unsigned short*>(table) + (size_t)id * C + c);
                v = make_float4(bf16_to_f32(h.x), bf16_to_f32(h.y), bf16_to_f32(h.z), bf16_to_f32(h.w));
            } else v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(table) + (size_t)id * C + c);
        }
        *reinterpret_cast<float4*>(out + (size_t)r * C + c) = v;
    }
}
void launch_gather_rows(const void* table, bool table_bf16, int64_t n_table, int C, const int64_t* ids, int rows, float* out,
                        int* err, hipStream_t st) {
    QTTS_REQUIRE(C % 4 == 0, QTTS_ERR_ARG, "gather: C % 4");
    if (table_bf16) hipLaunchKernelGGL(gather_rows_kernel<true>, dim3(rows), dim3(256), 0, st, table, n_table, C, ids, out, err);
    else hipLaunchKernelGGL(gather_rows_kernel<false>, dim3(rows), dim3(256), 0, st, table, n_table, C, ids, out, err);
    QTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace qtts
