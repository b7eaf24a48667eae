// TEST INFRASTRUCTURE -- CPU stand-ins for the launch interfaces csrc/talker_engine.hip uses (fp32 / exact mode only):
// the weight-streaming decode GEMM over its PACKED tile layout, fused q/k-norm + RoPE + paged-KV attention, the HF
// logits processors + greedy pick, and the device-resident loop-state glue.  Each restates the documented semantics of
// the HIP kernel it replaces (csrc/kernels.h, csrc/glue.h), so that the talker's real C++ -- weight packing in finalize(),
// prefill, the frame step, the stop bookkeeping, the C ABI -- runs in the CPU suite (tests/test_hostemu.py).
#include <cmath>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "glue.h"

namespace qtts {

// ---- skinny.hip host helpers (same layout: [N/fs strips][K/KT k-tiles][4 k-slices][fs features][16 B])
bool skinny_can_stage(int, int, bool bf16) {
    QTTS_REQUIRE(!bf16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    return false;
}
size_t skinny_packed_bytes(int N, int K, bool bf16) { return (size_t)N * K * (bf16 ? 2 : 4); }
void pack_skinny_weight(const float* W, int N, int K, bool bf16, void* out_host, const float* g, int fs) {
    QTTS_REQUIRE(!bf16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    const int KT = 16, nkt = K / KT, strips = N / fs;
    float* out = reinterpret_cast<float*>(out_host);
    for (int s = 0; s < strips; ++s)
        for (int kt = 0; kt < nkt; ++kt)
            for (int q = 0; q < 4; ++q)
                for (int i = 0; i < fs; ++i) {
                    const size_t tile = ((size_t)s * nkt + kt) * (fs * 4) + q * fs + i;
                    const int k0 = kt * KT + q * 4;
                    const float* src = W + (size_t)(s * fs + i) * K + k0;
                    for (int e = 0; e < 4; ++e) out[tile * 4 + e] = g ? src[e] * g[k0 + e] : src[e];
                }
}
static inline float packed_w(const float* Wp, int K, int fs, int n, int k) {       // inverse of the layout above
    const int KT = 16, nkt = K / KT;
    const int s = n / fs, i = n % fs, kt = k / KT, q = (k % KT) / 4, e = k % 4;
    return Wp[(((size_t)s * nkt + kt) * (fs * 4) + q * fs + i) * 4 + e];
}

// skinny.hip: out[m][n] = epi( rstd[m] * sum_k x[m][k] * W'[n][k] ), rstd from ss_in when `norm`; SwiGLU pairs 16-row strips
void launch_skinny(const SkinnyParams& p, bool bf16, hipStream_t) {
    QTTS_REQUIRE(!bf16 && !p.x_bf16 && !p.out_bf16 && !p.out16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    const int fs = p.fs ? p.fs : 16;
    QTTS_REQUIRE(fs == 16, QTTS_ERR_ARG, "skinny: narrow strips are only built for the staged bf16 kernel");
    QTTS_REQUIRE(p.N % 16 == 0 && p.K % 16 == 0 && p.M >= 1 && p.M <= 64, QTTS_ERR_ARG, "skinny: shape");
    QTTS_REQUIRE(!p.norm || p.ss_in, QTTS_ERR_ARG, "skinny: norm without LDS staging needs ss_in (row sums of squares)");
    if (p.done_flag && *p.done_flag) return;
    const float* Wp = reinterpret_cast<const float*>(p.Wp);
    std::vector<float> acc(p.N);
    for (int m = 0; m < p.M; ++m) {
        const float rstd = p.norm ? 1.f / sqrtf(p.ss_in[m] / (float)p.K + p.eps) : 1.f;
        for (int n = 0; n < p.N; ++n) {
            float s = 0.f;
            for (int k = 0; k < p.K; ++k) s += p.x[(size_t)m * p.ldx + k] * packed_w(Wp, p.K, fs, n, k);
            acc[n] = s * rstd + (p.bias ? p.bias[n] : 0.f);
        }
        if (p.act == ACT_SWIGLU) {
            for (int n = 0; n < p.N; ++n) {
                if ((n / 16) % 2) continue;
                const int col = (n / 32) * 16 + n % 16;
                float o = (acc[n] / (1.f + expf(-acc[n]))) * acc[n + 16];
                if (p.res) o += p.res[(size_t)m * p.ldr + col];
                p.out[(size_t)m * p.ldo + col] = o;
            }
        } else {
            for (int n = 0; n < p.N; ++n)
                p.out[(size_t)m * p.ldo + n] = acc[n] + (p.res ? p.res[(size_t)m * p.ldr + n] : 0.f);
        }
    }
}

// ---- glue (sampling.hip)
void launch_row_ss(const float* x, int ldx, int rows, int C, float* ss, const int* done, hipStream_t) {
    if (done && *done) return;
    for (int r = 0; r < rows; ++r) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += x[(size_t)r * ldx + c] * x[(size_t)r * ldx + c];
        ss[r] = s;
    }
}
void launch_apply_norm(const float* x, int ldx, const float* g, float eps, float* y, int ldy, int rows, int C, const int* done,
                       hipStream_t) {
    if (done && *done) return;
    for (int r = 0; r < rows; ++r) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += x[(size_t)r * ldx + c] * x[(size_t)r * ldx + c];
        const float rs = 1.f / sqrtf(s / (float)C + eps);
        for (int c = 0; c < C; ++c) y[(size_t)r * ldy + c] = g[c] * (x[(size_t)r * ldx + c] * rs);
    }
}
void launch_cp_gather(const CpGatherParams& p, hipStream_t) {
    if (p.done && *p.done) return;
    QTTS_REQUIRE(!p.out16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    const int rows = p.pass == 0 ? 2 * p.B : p.B;
    for (int r = 0; r < rows; ++r) {
        const float* src;
        if (p.pass == 0) {
            const int t = r / p.B, b = r % p.B;
            src = t == 0 ? p.past_hidden + (size_t)b * p.H : p.talker_emb + (size_t)p.cur_tok[b] * p.H;
        } else src = p.cp_emb + ((size_t)(p.pass - 1) * p.cp_vocab + p.sub[(size_t)r * p.sub_stride + p.pass - 1]) * p.H;
        memcpy(p.out + (size_t)r * p.H, src, (size_t)p.H * 4);
    }
}
void launch_embed_sum(const EmbedSumParams& p, hipStream_t) {
    if (*p.st.done) return;
    QTTS_REQUIRE(!p.x_out16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    const int f = *p.st.gen_step;
    for (int b = 0; b < p.B; ++b) {
        const int tok0 = p.cur_tok[b];
        for (int c = 0; c < p.H; ++c) {
            float a = p.talker_emb[(size_t)tok0 * p.H + c];
            for (int i = 0; i < p.G - 1; ++i) a += p.cp_emb[((size_t)i * p.cp_vocab + p.sub[(size_t)b * p.sub_stride + i]) * p.H + c];
            const float* tp = f < p.Tt ? p.trailing + ((size_t)b * p.Tt + f) * p.H : p.tts_pad;
            a += tp[c];
            p.x_out[(size_t)b * p.H + c] = a;
            if (p.hidden_out) p.hidden_out[((size_t)b * p.max_frames + f) * p.H + c] = p.past_hidden[(size_t)b * p.H + c];
        }
        if (f < p.max_frames)
            for (int g = 0; g < p.G; ++g)
                p.codes_out[((size_t)b * p.max_frames + f) * p.G + g] = g == 0 ? tok0 : p.sub[(size_t)b * p.sub_stride + g - 1];
    }
}
void launch_sample_finish(const StepState& st, int B, int max_new_tokens, hipStream_t) {
    if (*st.done) return;
    const int n = *st.n_generated + 1;
    *st.n_generated = n;
    *st.gen_step += 1;
    *st.kv_len += 1;
    int any = 0;
    for (int b = 0; b < B; ++b) any |= st.unfinished[b];
    if (n >= max_new_tokens || !any) { *st.done = 1; *st.final_count = n; }
}
void launch_assemble_rows(const AssembleParams& p, hipStream_t) {
    for (int r = 0; r < p.rows; ++r) {
        const int tr = p.desc[r * 4 + 0], cid = p.desc[r * 4 + 1], sr = p.desc[r * 4 + 2], rf = p.desc[r * 4 + 3];
        bool bad = tr >= p.proj_rows || cid >= p.vocab || sr >= p.n_spk || rf >= p.n_ref;
        if (rf >= 0 && !bad) {
            bad = p.ref_codes[(size_t)rf * p.G] < 0 || p.ref_codes[(size_t)rf * p.G] >= p.vocab;
            for (int g = 1; g < p.G; ++g) bad = bad || p.ref_codes[(size_t)rf * p.G + g] < 0 || p.ref_codes[(size_t)rf * p.G + g] >= p.cp_vocab;
        }
        if (bad) *p.err = 1;
        for (int c = 0; c < p.H; ++c) {
            float a = 0.f; bool have = false;
            if (!bad) {
                if (cid >= 0) { a = p.talker_emb[(size_t)cid * p.H + c]; have = true; }
                if (sr >= 0) { const float e = p.spk[(size_t)sr * p.H + c]; a = have ? a + e : e; have = true; }
                if (rf >= 0) {
                    float e = p.talker_emb[(size_t)p.ref_codes[(size_t)rf * p.G] * p.H + c];
                    for (int g = 1; g < p.G; ++g) e += p.cp_emb[((size_t)(g - 1) * p.cp_vocab + p.ref_codes[(size_t)rf * p.G + g]) * p.H + c];
                    a = have ? a + e : e; have = true;
                }
                if (tr >= 0) { const float t = p.proj[(size_t)tr * p.H + c]; a = have ? t + a : t; }
            }
            p.out[(size_t)r * p.H + c] = a;
        }
    }
}
void launch_gather_rows(const void* table, bool table_bf16, int64_t n_table, int C, const int64_t* ids, int rows, float* out,
                        int* err, hipStream_t) {
    QTTS_REQUIRE(!table_bf16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    for (int r = 0; r < rows; ++r) {
        const int64_t id = ids[r];
        const bool bad = id < 0 || id >= n_table;
        if (bad) *err = 1;
        for (int c = 0; c < C; ++c) out[(size_t)r * C + c] = bad ? 0.f : reinterpret_cast<const float*>(table)[(size_t)id * C + c];
    }
}

// ---- attention.hip (paged KV; fp32 cache)
static inline size_t kv_off(const KvCache& c, int layer, int b, int s, int kvh) {
    const int page = c.page_table[b * c.pages_per_seq + (s >> 4)];
    return ((((size_t)layer * c.n_pages + page) * c.nkv + kvh) * 16 + (s & 15)) * c.hd;
}
static void norm_rope(const float* v, const float* w, float eps, const float* inv_freq, float pos, int hd, float* out) {
    float ss = 0.f;
    for (int d = 0; d < hd; ++d) ss += v[d] * v[d];
    const float r = 1.f / sqrtf(ss / (float)hd + eps);
    const int half = hd / 2;
    for (int d = 0; d < half; ++d) {
        const float x0 = w[d] * (v[d] * r), x1 = w[d + half] * (v[d + half] * r);
        const float ang = pos * inv_freq[d], c = cosf(ang), s = sinf(ang);
        out[d] = x0 * c - x1 * s;
        out[d + half] = x1 * c + x0 * s;
    }
}
void launch_qknorm_rope_store(const QkNormRopeParams& p, hipStream_t) {
    QTTS_REQUIRE(!p.kv.bf16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    float* kc = reinterpret_cast<float*>(p.kv.k);
    float* vc = reinterpret_cast<float*>(p.kv.v);
    std::vector<float> tmp(p.hd);
    for (int b = 0; b < p.B; ++b)
        for (int t = p.n_pad[b]; t < p.T; ++t) {
            float* row = p.qkv + ((size_t)b * p.T + t) * p.ld;
            const float pos = (float)(t - p.n_pad[b]);
            for (int h = 0; h < p.nh + p.nkv; ++h) {
                float* v = row + h * p.hd;
                norm_rope(v, h < p.nh ? p.qw : p.kw, p.eps, p.inv_freq, pos, p.hd, tmp.data());
                memcpy(v, tmp.data(), (size_t)p.hd * 4);
                if (h >= p.nh) memcpy(kc + kv_off(p.kv, p.layer, b, t, h - p.nh), tmp.data(), (size_t)p.hd * 4);
            }
            for (int kvh = 0; kvh < p.nkv; ++kvh)
                memcpy(vc + kv_off(p.kv, p.layer, b, t, kvh), row + (p.nh + p.nkv + kvh) * p.hd, (size_t)p.hd * 4);
        }
}
void launch_attn_decode(const AttnDecodeParams& p, hipStream_t) {
    QTTS_REQUIRE(p.hd == 128 && !p.kv.bf16 && !p.out_bf16, QTTS_ERR_ARG, "host emulation: head_dim 128, fp32 cache");
    if (p.done_flag && *p.done_flag) return;
    float* kc = reinterpret_cast<float*>(p.kv.k);
    float* vc = reinterpret_cast<float*>(p.kv.v);
    const int S0 = p.len_dev ? *p.len_dev : p.len_static, GQ = p.nh / p.nkv, hd = p.hd;
    const float scale = 1.f / sqrtf((float)hd);
    std::vector<float> q(hd), k(hd), sc;
    for (int b = 0; b < p.B; ++b) {
        const int npad = p.n_pad ? p.n_pad[b] : 0;
        for (int t = 0; t < p.n_new; ++t) {                              // append the new keys / values first
            const float* row = p.qkv + ((size_t)t * p.B + b) * p.ld;
            for (int kvh = 0; kvh < p.nkv; ++kvh) {
                norm_rope(row + (p.nh + kvh) * hd, p.kw, p.eps, p.inv_freq, (float)(S0 + t - npad), hd, k.data());
                memcpy(kc + kv_off(p.kv, p.layer, b, S0 + t, kvh), k.data(), (size_t)hd * 4);
                memcpy(vc + kv_off(p.kv, p.layer, b, S0 + t, kvh), row + (p.nh + p.nkv + kvh) * hd, (size_t)hd * 4);
            }
        }
        for (int t = 0; t < p.n_new; ++t) {
            const float* row = p.qkv + ((size_t)t * p.B + b) * p.ld;
            const int hi = S0 + t;
            sc.assign(hi + 1, 0.f);
            for (int h = 0; h < p.nh; ++h) {
                const int kvh = h / GQ;
                norm_rope(row + h * hd, p.qw, p.eps, p.inv_freq, (float)(S0 + t - npad), hd, q.data());
                float m = -INFINITY;
                for (int s = npad; s <= hi; ++s) {
                    const float* kk = kc + kv_off(p.kv, p.layer, b, s, kvh);
                    float d = 0.f;
                    for (int e = 0; e < hd; ++e) d += q[e] * kk[e];
                    sc[s] = d * scale; m = fmaxf(m, sc[s]);
                }
                float l = 0.f;
                float* o = p.out + ((size_t)t * p.B + b) * p.ldo + h * hd;
                for (int e = 0; e < hd; ++e) o[e] = 0.f;
                for (int s = npad; s <= hi; ++s) {
                    const float pr = expf(sc[s] - m);
                    l += pr;
                    const float* vv = vc + kv_off(p.kv, p.layer, b, s, kvh);
                    for (int e = 0; e < hd; ++e) o[e] += pr * vv[e];
                }
                for (int e = 0; e < hd; ++e) o[e] /= l;
            }
        }
    }
}

// ---- sampling.hip: HF processors (RepetitionPenalty -> MinNewTokens -> SuppressTokens) + greedy pick + bookkeeping
void launch_sample(const SampleParams& p, hipStream_t) {
    QTTS_REQUIRE(!p.do_sample, QTTS_ERR_ARG, "host emulation implements the greedy sampler only");
    if (p.done_in && *p.done_in) return;
    std::vector<float> sc(p.V);
    const int n_gen = p.n_generated_dev ? *p.n_generated_dev : 0;
    for (int b = 0; b < p.B; ++b) {
        const float* lg = p.logits + (size_t)b * p.ld;
        for (int v = 0; v < p.V; ++v) sc[v] = lg[v];
        if (p.generated && p.repetition_penalty != 1.0f)
            for (int i = 0; i < n_gen; ++i) {
                const int tok = p.generated[(size_t)b * p.gen_stride + i];
                sc[tok] = lg[tok] < 0.f ? lg[tok] * p.repetition_penalty : lg[tok] / p.repetition_penalty;
            }
        if (p.eos >= 0 && n_gen < p.min_new_tokens) sc[p.eos] = -INFINITY;
        if (p.suppress_mask)
            for (int v = 0; v < p.V; ++v) if (p.suppress_mask[v]) sc[v] = -INFINITY;
        int token = 0; float bv = -INFINITY;
        for (int v = 0; v < p.V; ++v) if (sc[v] > bv) { bv = sc[v]; token = v; }      // lowest index wins ties
        if (p.gather_emb) {
            QTTS_REQUIRE(!p.gather_out16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
            memcpy(p.gather_out + (size_t)b * p.gather_C, p.gather_emb + (size_t)token * p.gather_C, (size_t)p.gather_C * 4);
        }
        if (p.unfinished) {
            const int uf = p.unfinished[b];
            if (!uf) token = p.eos;
            p.unfinished[b] = uf && (token != p.eos);
            if (p.generated_out) p.generated_out[(size_t)b * p.gen_stride + n_gen] = token;
        }
        p.tok_out[(size_t)b * p.tok_stride] = token;
    }
}

}  // namespace qtts
