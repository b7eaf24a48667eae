// encoder_engine.hip -- Qwen3-TTS-Tokenizer-12Hz ENCODER (waveform -> codes) on gfx950: SURVEY.md 8(f3).
//
// The reference's encoder is transformers.MimiModel.encode behind Qwen3TTSTokenizerV2Model.encode (tokenizer
// v2:897-908, 961-991); algorithm and parity anchor in oracle/codec_enc_ref.py.  Layout: channel-last fp32 rows, as in the
// decoder.  Mapping onto the existing kernels (mirrored op by op in oracle/codec_enc_stage_emul.py):
//   * stride-1 causal convs (k = 7 / 3 / 1, dilation)  -> gemm_tap taps with zero left padding
//   * strided convs (k = 2r, stride r)                 -> view r consecutive rows as one "super-row" of r*C channels (free
//     in channel-last layout, after zero-padding the length to a multiple of r == MimiConv1d's extra padding): then
//     out[t] = W[:, :, :r] . super[t-1] + W[:, :, r:] . super[t], a 2-tap causal GEMM with K = r*C
//   * downsample (k = 4, stride 2, REPLICATE padding)  -> pad_rows(replicate) by one super-row on the left, same GEMM, drop
//     the first output row
//   * transformer: LayerNorm, fused qkv GEMM, rope_inplace, attn_rows(sliding window), GELU MLP, LayerScale epilogues
//   * split RVQ: scores = r . E^T on the exact-fp32 GEMM, argmin(||e||^2 - 2 r.e) + residual update per layer
// STATUS round 1: compiled for gfx950; hardware run pending (tests gated behind QTTS_EXPERIMENTAL=1).
#include <map>
#include <algorithm>
#include "common.h"
#include "kernels.h"

using namespace qtts;

namespace {
struct ELin {                 // one gemm_tap operator (weights fp32 or bf16 per engine mode; `f32` forces fp32)
    DevBuf W, bias;
    int N = 0, K = 0, taps = 1;
    int shift[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool has_bias = false, f32 = false;
};
}  // namespace

struct qtts_encoder {
    qtts_encoder_config cfg;
    bool bf16 = false, finalized = false;
    std::map<std::string, std::vector<float>> host;
    std::map<std::string, std::vector<int64_t>> shapes;

    DevBuf w0, b0;                       // first conv (C_in = 1)
    int c0 = 0;
    struct Res { ELin c1, c2; };
    struct Stage { std::vector<Res> res; ELin down; int ratio = 1, cin = 0; };
    std::vector<Stage> stages;
    ELin last;
    struct TL { ELin qkv, o, fc1, fc2; DevBuf n1w, n1b, n2w, n2b, ls1, ls2; };
    std::vector<TL> tl;
    DevBuf inv_freq;
    ELin downsample;
    struct VQ { ELin in_proj; std::vector<ELin> score; std::vector<DevBuf> table, enorm; };
    VQ sem, aco;
    DevBuf buf[4];
    size_t buf_elems = 0;

    std::vector<float>& P(const std::string& n, std::initializer_list<int64_t> want) {
        auto it = host.find(n);
        if (it == host.end()) throw Error(QTTS_ERR_UNBOUND, "encoder weight not bound: " + n);
        auto& s = shapes[n];
        if (s.size() != want.size() || !std::equal(s.begin(), s.end(), want.begin())) {
            std::string a, b;
            for (auto d : s) a += std::to_string(d) + ",";
            for (auto d : want) b += std::to_string(d) + ",";
            throw Error(QTTS_ERR_ARG, "encoder weight " + n + " has shape (" + a + ") but the config implies (" + b + ")");
        }
        return it->second;
    }
    void upload_w(ELin& l, const std::vector<float>& w) {
        if (bf16 && !l.f32) {
            std::vector<bf16_t> h(w.size());
            for (size_t i = 0; i < w.size(); ++i) h[i] = f32_to_bf16(w[i]);
            l.W.upload(h.data(), h.size() * 2);
        } else l.W.upload(w.data(), w.size() * 4);
    }
    static void upload_f(DevBuf& d, const std::vector<float>& w) { d.upload(w.data(), w.size() * 4); }

    // stride-1 causal Conv1d (Cout, Cin, k), dilation d  -> [tap][Cout][Cin], shift_j = -(k-1-j)*d
    void make_conv(ELin& l, const std::string& pfx, int Co, int Ci, int k, int dil) {
        auto& w = P(pfx + ".weight", {Co, Ci, k});
        QTTS_REQUIRE(k <= 8 && Ci % 32 == 0, QTTS_ERR_ARG, "encoder conv: kernel <= 8 and in-channels % 32 (" + pfx + ")");
        std::vector<float> r((size_t)k * Co * Ci);
        for (int j = 0; j < k; ++j)
            for (int n = 0; n < Co; ++n)
                for (int c = 0; c < Ci; ++c) r[((size_t)j * Co + n) * Ci + c] = w[((size_t)n * Ci + c) * k + j];
        l.N = Co; l.K = Ci; l.taps = k;
        for (int j = 0; j < k; ++j) l.shift[j] = -(k - 1 - j) * dil;
        upload_w(l, r);
        upload_f(l.bias, P(pfx + ".bias", {Co})); l.has_bias = true;
    }
    // strided Conv1d (Cout, Cin, 2r), stride r -> 2 taps over super-rows: W_tap[n][jj*Cin + ci] = w[n][ci][tap*r + jj]
    void make_strided(ELin& l, const std::string& pfx, int Co, int Ci, int r, bool bias) {
        auto& w = P(pfx + ".weight", {Co, Ci, 2 * r});
        QTTS_REQUIRE((r * Ci) % 32 == 0, QTTS_ERR_ARG, "encoder strided conv: stride * in-channels % 32 (" + pfx + ")");
        std::vector<float> rp((size_t)2 * Co * r * Ci);
        for (int t = 0; t < 2; ++t)
            for (int n = 0; n < Co; ++n)
                for (int jj = 0; jj < r; ++jj)
                    for (int ci = 0; ci < Ci; ++ci)
                        rp[(((size_t)t * Co + n) * r + jj) * Ci + ci] = w[((size_t)n * Ci + ci) * (2 * r) + t * r + jj];
        l.N = Co; l.K = r * Ci; l.taps = 2; l.shift[0] = -1; l.shift[1] = 0;
        upload_w(l, rp);
        if (bias) { upload_f(l.bias, P(pfx + ".bias", {Co})); l.has_bias = true; }
    }
    void make_linear(ELin& l, const std::vector<float>& w, int N, int K) {
        QTTS_REQUIRE(K % 32 == 0, QTTS_ERR_ARG, "encoder linear: in-features % 32");
        l.N = N; l.K = K; l.taps = 1; l.shift[0] = 0;
        upload_w(l, w);
    }
    void gemm(const ELin& l, const float* A, int lda, int M, int T, float* C, int ldc, int act, const float* res, int ldr,
              const float* scale, hipStream_t st) {
        GemmTapParams p{};
        p.A = A; p.lda = lda; p.M = M; p.T = T; p.W = l.W.p; p.N = l.N; p.K = l.K; p.taps = l.taps;
        for (int i = 0; i < 8; ++i) p.shift[i] = l.shift[i];
        p.bias = l.has_bias ? l.bias.as<float>() : nullptr;
        p.scale = scale; p.res = res; p.ldr = ldr; p.snake_ea = nullptr; p.snake_ib = nullptr;
        p.act = act; p.C = C; p.ldc = ldc;
        launch_gemm_tap(p, bf16 && !l.f32, st);
    }
    void make_vq(VQ& v, const std::string& pfx, int n_layers) {
        const auto& c = cfg;
        auto& ip = P(pfx + "input_proj.weight", {c.codebook_dim, c.hidden_size, 1});
        v.in_proj.f32 = true;                                  // the quantiser runs in fp32 in either engine mode
        make_linear(v.in_proj, ip, c.codebook_dim, c.hidden_size);
        v.score.resize(n_layers); v.table.resize(n_layers); v.enorm.resize(n_layers);
        for (int i = 0; i < n_layers; ++i) {
            const std::string p = pfx + "layers." + std::to_string(i) + ".codebook.";
            auto& es = P(p + "embed_sum", {c.codebook_size, c.codebook_dim});
            auto& cu = P(p + "cluster_usage", {c.codebook_size});
            std::vector<float> t((size_t)c.codebook_size * c.codebook_dim), en(c.codebook_size);
            for (int j = 0; j < c.codebook_size; ++j) {
                const float d = std::max(cu[j], 1e-5f);                   // embed_sum / clamp(cluster_usage, eps)
                float q = 0.f;
                for (int e = 0; e < c.codebook_dim; ++e) {
                    const float x = es[(size_t)j * c.codebook_dim + e] / d;
                    t[(size_t)j * c.codebook_dim + e] = x;
                    q += x * x;
                }
                en[j] = q;
            }
            v.score[i].f32 = true;
            make_linear(v.score[i], t, c.codebook_size, c.codebook_dim);
            upload_f(v.table[i], t);
            upload_f(v.enorm[i], en);
        }
    }
    int64_t frames_for(int64_t samples) const {
        int64_t T = samples;
        for (auto& s : stages) T = (T + s.ratio - 1) / s.ratio;
        return (T + 1) / 2;
    }
    void finalize();
    void encode(const float* wav, int B, int L, int64_t* codes, hipStream_t st);
};

void qtts_encoder::finalize() {
    const auto& c = cfg;
    QTTS_REQUIRE(c.head_dim == 64 || c.head_dim == 128, QTTS_ERR_ARG, "encoder head_dim must be 64 or 128");
    QTTS_REQUIRE(c.n_ratios >= 1 && c.n_ratios <= 8, QTTS_ERR_ARG, "encoder: 1..8 ratios");
    QTTS_REQUIRE(c.kernel_size <= 8, QTTS_ERR_ARG, "encoder: first kernel <= 8");
    QTTS_REQUIRE(c.num_semantic_quantizers >= 1 && c.valid_num_quantizers > c.num_semantic_quantizers &&
                     c.valid_num_quantizers <= c.num_quantizers, QTTS_ERR_ARG, "encoder: quantizer counts");
    c0 = c.num_filters;
    {
        auto& w = P("encoder.layers.0.conv.weight", {c0, 1, c.kernel_size});
        upload_f(w0, w);
        upload_f(b0, P("encoder.layers.0.conv.bias", {c0}));
    }
    int idx = 1, ch = c0;
    stages.resize(c.n_ratios);
    for (int s = 0; s < c.n_ratios; ++s) {
        const int ratio = c.ratios[c.n_ratios - 1 - s];                 // reversed(upsampling_ratios)
        auto& S = stages[s];
        S.ratio = ratio; S.cin = ch;
        S.res.resize(c.num_residual_layers);
        int dil = 1;
        for (int j = 0; j < c.num_residual_layers; ++j) {
            const std::string p = "encoder.layers." + std::to_string(idx) + ".block.";
            make_conv(S.res[j].c1, p + "1.conv", ch / c.compress, ch, c.residual_kernel_size, dil);
            make_conv(S.res[j].c2, p + "3.conv", ch, ch / c.compress, 1, 1);
            dil *= c.dilation_growth_rate;
            ++idx;
        }
        ++idx;                                                          // the ELU module
        make_strided(S.down, "encoder.layers." + std::to_string(idx) + ".conv", ch * 2, ch, ratio, true);
        ++idx;
        ch *= 2;
    }
    ++idx;                                                              // ELU
    make_conv(last, "encoder.layers." + std::to_string(idx) + ".conv", c.hidden_size, ch, c.last_kernel_size, 1);
    const int H = c.hidden_size, I = c.intermediate_size;
    const int qd = c.num_attention_heads * c.head_dim, kvd = c.num_key_value_heads * c.head_dim;
    tl.resize(c.num_hidden_layers);
    for (int l = 0; l < c.num_hidden_layers; ++l) {
        const std::string p = "encoder_transformer.layers." + std::to_string(l) + ".";
        auto& L = tl[l];
        auto& q = P(p + "self_attn.q_proj.weight", {qd, H});
        auto& k = P(p + "self_attn.k_proj.weight", {kvd, H});
        auto& v = P(p + "self_attn.v_proj.weight", {kvd, H});
        std::vector<float> w; w.reserve(q.size() + k.size() + v.size());
        w.insert(w.end(), q.begin(), q.end()); w.insert(w.end(), k.begin(), k.end()); w.insert(w.end(), v.begin(), v.end());
        make_linear(L.qkv, w, qd + 2 * kvd, H);
        make_linear(L.o, P(p + "self_attn.o_proj.weight", {H, qd}), H, qd);
        make_linear(L.fc1, P(p + "mlp.fc1.weight", {I, H}), I, H);
        make_linear(L.fc2, P(p + "mlp.fc2.weight", {H, I}), H, I);
        upload_f(L.n1w, P(p + "input_layernorm.weight", {H})); upload_f(L.n1b, P(p + "input_layernorm.bias", {H}));
        upload_f(L.n2w, P(p + "post_attention_layernorm.weight", {H})); upload_f(L.n2b, P(p + "post_attention_layernorm.bias", {H}));
        upload_f(L.ls1, P(p + "self_attn_layer_scale.scale", {H})); upload_f(L.ls2, P(p + "mlp_layer_scale.scale", {H}));
    }
    {
        std::vector<float> f(c.head_dim / 2);
        for (int i = 0; i < c.head_dim / 2; ++i) f[i] = 1.0f / powf(c.rope_theta, (float)(2 * i) / (float)c.head_dim);
        upload_f(inv_freq, f);
    }
    make_strided(downsample, "downsample.conv", H, H, 2, false);
    make_vq(sem, "quantizer.semantic_residual_vector_quantizer.", c.num_semantic_quantizers);
    make_vq(aco, "quantizer.acoustic_residual_vector_quantizer.", c.valid_num_quantizers - c.num_semantic_quantizers);
    // workspace: the largest rows x channels of any stage for max_samples (+ one super-row of padding), per sequence
    size_t per_seq = 0;
    {
        int64_t T = c.max_samples; int C = c0;
        per_seq = (size_t)(T + 8) * C;
        for (auto& s : stages) {
            per_seq = std::max(per_seq, (size_t)(T + s.ratio) * C);
            T = (T + s.ratio - 1) / s.ratio; C *= 2;
            per_seq = std::max(per_seq, (size_t)T * C);
        }
        per_seq = std::max(per_seq, (size_t)(T + 4) * std::max({qd + 2 * kvd, I, H, c.codebook_size}));
    }
    buf_elems = per_seq * (size_t)std::max(1, c.max_batch);
    for (auto& b : buf) b.alloc(buf_elems * sizeof(float));
    host.clear();
    finalized = true;
}

void qtts_encoder::encode(const float* wav, int B, int L, int64_t* codes, hipStream_t st) {
    const auto& c = cfg;
    QTTS_REQUIRE(finalized, QTTS_ERR_STATE, "encoder: finalize() first");
    QTTS_REQUIRE(B >= 1 && B <= c.max_batch && L >= 1 && L <= c.max_samples, QTTS_ERR_LIMIT,
                 "encoder: batch / samples exceed max_batch / max_samples given at create");
    float* pool[4] = {buf[0].as<float>(), buf[1].as<float>(), buf[2].as<float>(), buf[3].as<float>()};
    auto other = [&](std::initializer_list<const float*> busy) -> float* {
        for (float* p : pool) {
            bool used = false;
            for (const float* q : busy) used = used || q == p;
            if (!used) return p;
        }
        throw Error(QTTS_ERR_STATE, "encoder: out of workspace buffers");
    };
    auto fits = [&](int64_t rows, int C) {
        QTTS_REQUIRE((size_t)rows * (size_t)C <= buf_elems, QTTS_ERR_LIMIT, "encoder: workspace too small");
    };
    // ---- SEANet encoder (TM:450-492)
    float* x = pool[0];
    int T = L, C = c0;
    fits((int64_t)B * T, C);
    launch_conv_in1(wav, w0.as<float>(), b0.as<float>(), x, B, L, C, c.kernel_size, st);
    for (auto& S : stages) {
        for (auto& R : S.res) {                                          // x += conv1(elu(conv3(elu(x))))
            float* a = other({x});
            float* h = other({x, a});
            launch_elu(x, a, (int64_t)B * T * C, st);
            gemm(R.c1, a, C, B * T, T, h, C / c.compress, ACT_NONE, nullptr, 0, nullptr, st);
            launch_elu(h, a, (int64_t)B * T * (C / c.compress), st);
            gemm(R.c2, a, C / c.compress, B * T, T, x, C, ACT_NONE, x, C, nullptr, st);
        }
        // ELU, zero-pad the length to a multiple of the stride, 2-tap GEMM over super-rows
        const int r = S.ratio, extra = (r - T % r) % r, Ts = (T + extra) / r;
        float* a = other({x});
        launch_elu(x, a, (int64_t)B * T * C, st);
        const float* sup = a;
        if (extra) {
            float* pd = other({x, a});
            fits((int64_t)B * (T + extra), C);
            launch_pad_rows(a, T, 0, extra, 0, pd, B, C, st);
            sup = pd;
        }
        float* y = other({sup, a});
        fits((int64_t)B * Ts, 2 * C);
        gemm(S.down, sup, r * C, B * Ts, Ts, y, 2 * C, ACT_NONE, nullptr, 0, nullptr, st);
        x = y; T = Ts; C *= 2;
    }
    {
        float* a = other({x});
        float* y = other({x, a});
        launch_elu(x, a, (int64_t)B * T * C, st);
        gemm(last, a, C, B * T, T, y, c.hidden_size, ACT_NONE, nullptr, 0, nullptr, st);
        x = y; C = c.hidden_size;
    }
    // ---- encoder transformer (TM:782-928): pre-LN, RoPE, sliding-window causal attention, GELU MLP, LayerScale
    {
        const int H = c.hidden_size, I = c.intermediate_size, M = B * T;
        const int qd = c.num_attention_heads * c.head_dim, kvd = c.num_key_value_heads * c.head_dim, qw = qd + 2 * kvd;
        fits((int64_t)M, std::max({qw, I, H}));
        float* h = x;
        float* a = other({h});
        float* b2 = other({h, a});
        for (auto& Ly : tl) {
            launch_layernorm(h, H, Ly.n1w.as<float>(), Ly.n1b.as<float>(), c.norm_eps, a, H, M, H, st);
            gemm(Ly.qkv, a, H, M, T, b2, qw, ACT_NONE, nullptr, 0, nullptr, st);
            launch_rope_inplace(b2, qw, M, T, c.num_attention_heads + c.num_key_value_heads, c.head_dim, inv_freq.as<float>(), st);
            AttnRowsParams ap{};
            ap.qkv = b2; ap.ld = qw; ap.q_off = 0; ap.k_off = qd; ap.v_off = qd + kvd;
            ap.B = B; ap.T = T; ap.nh = c.num_attention_heads; ap.nkv = c.num_key_value_heads; ap.hd = c.head_dim;
            ap.window = c.sliding_window; ap.n_pad = nullptr; ap.out = a; ap.ldo = qd;
            launch_attn_rows(ap, st);
            gemm(Ly.o, a, qd, M, T, h, H, ACT_NONE, h, H, Ly.ls1.as<float>(), st);
            launch_layernorm(h, H, Ly.n2w.as<float>(), Ly.n2b.as<float>(), c.norm_eps, a, H, M, H, st);
            gemm(Ly.fc1, a, H, M, T, b2, I, ACT_GELU, nullptr, 0, nullptr, st);
            gemm(Ly.fc2, b2, I, M, T, h, H, ACT_NONE, h, H, Ly.ls2.as<float>(), st);
        }
        x = h;
    }
    // ---- downsample: k = 4, stride 2, replicate padding (2 rows left, to an even length right) (TM:1197-1207)
    {
        const int H = c.hidden_size, extra = T % 2, Tp = 2 + T + extra, Ts = Tp / 2;
        float* pd = other({x});
        fits((int64_t)B * Tp, H);
        launch_pad_rows(x, T, 2, extra, 1, pd, B, H, st);
        float* y = other({x, pd});
        gemm(downsample, pd, 2 * H, B * Ts, Ts, y, H, ACT_NONE, nullptr, 0, nullptr, st);
        float* z = other({y, pd});
        launch_stage_rows(y, Ts, 1, Ts - 1, nullptr, 0, z, B, H, st);    // drop the row that belongs to the padding
        x = z; T = Ts - 1; C = H;
    }
    // ---- split residual VQ (TM:1050-1123): semantic and acoustic quantisers both start from the same embeddings
    {
        const int D = c.codebook_dim, nq = c.valid_num_quantizers, ns = c.num_semantic_quantizers;
        const int64_t stride_b = (int64_t)nq * T;
        float* r = other({x});
        float* sc = other({x, r});
        fits((int64_t)B * T, std::max(D, c.codebook_size));
        auto run = [&](VQ& v, int q0) {
            gemm(v.in_proj, x, C, B * T, T, r, D, ACT_NONE, nullptr, 0, nullptr, st);
            for (size_t i = 0; i < v.score.size(); ++i) {
                gemm(v.score[i], r, D, B * T, T, sc, c.codebook_size, ACT_NONE, nullptr, 0, nullptr, st);
                launch_vq_argmin_update(sc, c.codebook_size, v.enorm[i].as<float>(), v.table[i].as<float>(), D, r,
                                        codes + (int64_t)(q0 + (int)i) * T, stride_b, B, T, st);
            }
        };
        run(sem, 0);
        run(aco, ns);
    }
}

// ============================================================================================ C ABI
namespace qtts { void set_last_error(const std::string& s); }
#define QTTS_API_BEGIN try {
#define QTTS_API_END                                                        \
    }                                                                       \
    catch (const qtts::Error& e) { qtts::set_last_error(e.what()); return e.code; } \
    catch (const std::exception& e) { qtts::set_last_error(e.what()); return QTTS_ERR_ARG; } \
    return QTTS_OK;

extern "C" {

int qtts_encoder_create(const qtts_encoder_config* cfg, qtts_encoder** out) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(cfg && out, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(cfg->compute_dtype == QTTS_F32 || cfg->compute_dtype == QTTS_BF16, QTTS_ERR_ARG, "compute_dtype");
    int ndev = 0;
    if (!getenv("QTTS_DEBUG_NO_DEVICE")) {
        QTTS_CHECK_HIP(hipGetDeviceCount(&ndev));
        QTTS_REQUIRE(ndev > 0, QTTS_ERR_HIP, "no HIP device");
    }
    auto* e = new qtts_encoder();
    e->cfg = *cfg;
    e->bf16 = cfg->compute_dtype == QTTS_BF16;
    *out = e;
    QTTS_API_END
}
void qtts_encoder_destroy(qtts_encoder* e) { delete e; }
int qtts_encoder_bind(qtts_encoder* e, const char* name, const void* hostp, int32_t src_dtype, int32_t ndim, const int64_t* shape) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(e && name && hostp && shape, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(!e->finalized, QTTS_ERR_STATE, "bind after finalize");
    QTTS_REQUIRE(src_dtype == QTTS_F32 || src_dtype == QTTS_BF16, QTTS_ERR_ARG, "src_dtype");
    HostTensor ht{hostp, src_dtype, std::vector<int64_t>(shape, shape + ndim)};
    e->host[name] = ht.to_f32();
    e->shapes[name] = ht.shape;
    QTTS_API_END
}
int qtts_encoder_finalize(qtts_encoder* e) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(e, QTTS_ERR_ARG, "null handle");
    e->finalize();
    QTTS_API_END
}
int qtts_encoder_frames(qtts_encoder* e, int64_t samples, int64_t* frames) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(e && frames && samples >= 1, QTTS_ERR_ARG, "bad argument");
    QTTS_REQUIRE(e->finalized, QTTS_ERR_STATE, "encoder: finalize() first");
    *frames = e->frames_for(samples);
    QTTS_API_END
}
int qtts_encoder_encode(qtts_encoder* e, const float* wav_dev, int32_t B, int32_t samples, int64_t* codes_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(e && wav_dev && codes_dev, QTTS_ERR_ARG, "null argument");
    e->encode(wav_dev, B, samples, codes_dev, (hipStream_t)stream);
    QTTS_API_END
}

}  // extern "C"
