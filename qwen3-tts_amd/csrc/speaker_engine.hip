// speaker_engine.hip -- speaker embedding of the Base (voice-clone) model on gfx950: SURVEY.md 8(f4).
//
// Replaces Qwen3TTSForConditionalGeneration.extract_speaker_embedding (modeling_qwen3_tts.py:1941-1954):
// mel_spectrogram (M:402-464) -> Qwen3TTSSpeakerEncoder (ECAPA-TDNN, M:95-393).  Oracle: oracle/speaker_ref.py (the
// ECAPA part bit-identical to the reference module; the Slaney filterbank a restatement of librosa's, unpinned).
// Mapping (channel-last fp32 rows; mirrored in oracle/speaker_stage_emul.py):
//   * STFT: the reflect-padded waveform viewed as rows of `hop` samples; a frame is n_fft/hop consecutive rows, so the
//     Hann-windowed DFT is a (n_fft/hop)-tap GEMM with K = hop, N = 2*(n_fft/2+1) (exact-fp32 MFMA), then magnitude,
//     the mel projection (a GEMM) and log(clamp(., 1e-5))
//   * every TDNN / 1x1 conv: gemm_tap; "same" reflect padding is materialised by a small staging kernel and the GEMM's
//     causal taps then land on output row t + (k-1)*dilation, which the next copy drops (`skip`)
//   * Res2Net: channel slices are strided views (lda / ldd), the running sum is fused into the padding kernel
//   * squeeze-excitation, attentive statistics pooling: per-channel time statistics + tiny GEMMs
// STATUS round 1: compiled for gfx950; orchestration executed in the CPU suite on kernel stand-ins; no hardware run yet.
#include <map>
#include <algorithm>
#include "common.h"
#include "kernels.h"

using namespace qtts;

namespace {
struct SLin {
    DevBuf W, bias;
    int N = 0, K = 0, taps = 1, k = 1, dil = 1;
    int shift[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool has_bias = false, f32 = false;
};
}  // namespace

struct qtts_speaker {
    qtts_speaker_config cfg;
    bool bf16 = false, finalized = false;
    std::map<std::string, std::vector<float>> host;
    std::map<std::string, std::vector<int64_t>> shapes;

    SLin dft, melw;                      // front end
    int nb = 0, Kp = 0, ntap = 0;        // n_fft/2+1 bins, padded to Kp columns, taps per frame
    SLin tdnn0, mfa, asp_tdnn, asp_conv, fc;
    struct Blk { SLin tdnn1, tdnn2, se1, se2; std::vector<SLin> res; int dil = 1; };
    std::vector<Blk> blks;
    DevBuf buf[6];
    size_t buf_elems = 0;

    std::vector<float>& P(const std::string& n, std::initializer_list<int64_t> want) {
        auto it = host.find(n);
        if (it == host.end()) throw Error(QTTS_ERR_UNBOUND, "speaker weight not bound: " + n);
        auto& s = shapes[n];
        if (s.size() != want.size() || !std::equal(s.begin(), s.end(), want.begin())) {
            std::string a, b;
            for (auto d : s) a += std::to_string(d) + ",";
            for (auto d : want) b += std::to_string(d) + ",";
            throw Error(QTTS_ERR_ARG, "speaker weight " + n + " has shape (" + a + ") but the config implies (" + b + ")");
        }
        return it->second;
    }
    void upload_w(SLin& l, const std::vector<float>& w) {
        if (bf16 && !l.f32) {
            std::vector<bf16_t> h(w.size());
            for (size_t i = 0; i < w.size(); ++i) h[i] = f32_to_bf16(w[i]);
            l.W.upload(h.data(), h.size() * 2);
        } else l.W.upload(w.data(), w.size() * 4);
    }
    // nn.Conv1d (Cout, Cin, k), dilation d, padding "same" reflect: causal taps over the reflect-padded rows
    void make_conv(SLin& l, const std::string& pfx, int Co, int Ci, int k, int dil) {
        auto& w = P(pfx + ".weight", {Co, Ci, k});
        QTTS_REQUIRE(k >= 1 && k <= 8 && k % 2 == 1, QTTS_ERR_ARG, "speaker conv: odd kernel <= 7 (" + pfx + ")");
        QTTS_REQUIRE(Ci % 32 == 0, QTTS_ERR_ARG, "speaker conv: in-channels % 32 (" + pfx + ")");
        std::vector<float> r((size_t)k * Co * Ci);
        for (int j = 0; j < k; ++j)
            for (int n = 0; n < Co; ++n)
                for (int c = 0; c < Ci; ++c) r[((size_t)j * Co + n) * Ci + c] = w[((size_t)n * Ci + c) * k + j];
        l.N = Co; l.K = Ci; l.taps = k; l.k = k; l.dil = dil;
        for (int j = 0; j < k; ++j) l.shift[j] = -(k - 1 - j) * dil;
        upload_w(l, r);
        l.bias.upload(P(pfx + ".bias", {Co}).data(), (size_t)Co * 4); l.has_bias = true;
    }
    void gemm(const SLin& l, const float* A, int lda, int M, int T, float* C, int ldc, hipStream_t st) {
        GemmTapParams p{};
        p.A = A; p.lda = lda; p.M = M; p.T = T; p.W = l.W.p; p.N = l.N; p.K = l.K; p.taps = l.taps;
        for (int i = 0; i < 8; ++i) p.shift[i] = l.shift[i];
        p.bias = l.has_bias ? l.bias.as<float>() : nullptr;
        p.scale = nullptr; p.res = nullptr; p.ldr = 0; p.snake_ea = nullptr; p.snake_ib = nullptr;
        p.act = ACT_NONE; p.C = C; p.ldc = ldc;
        launch_gemm_tap(p, bf16 && !l.f32, st);
    }
    int64_t mel_frames(int64_t samples) const {
        const int pad = (cfg.n_fft - cfg.hop_size) / 2;
        return (samples + 2 * pad) / cfg.hop_size - (ntap - 1);
    }
    void finalize();
    void embed(const float* wav, int B, int S, float* out, float* mels_out, hipStream_t st);
    // y[dst] = act(conv(x)) with reflect "same" padding; x = src1 (+ src2), both strided row views of C = l.K channels
    void tdnn(const SLin& l, const float* src1, int ld1, const float* src2, int ld2, int B, int T, int act, float* dst, int ldd,
              float* scratch_pad, float* scratch_out, hipStream_t st) {
        const int total = l.dil * (l.k - 1), p = total / 2;
        const float* a = src1; int lda = ld1; int Tp = T;
        if (p > 0 || src2) {
            launch_reflect_pad_add_rows(src1, ld1, src2, ld2, T, p, l.K, scratch_pad, B, st);
            a = scratch_pad; lda = l.K; Tp = T + 2 * p;
        }
        gemm(l, a, lda, B * Tp, Tp, scratch_out, l.N, st);
        launch_copy_act_rows(scratch_out, l.N, Tp, total, T, l.N, act, dst, ldd, B, st);
    }
};

void qtts_speaker::finalize() {
    const auto& c = cfg;
    QTTS_REQUIRE(c.n_blocks >= 3 && c.n_blocks <= 8, QTTS_ERR_ARG, "speaker: 3..8 enc_channels entries");
    QTTS_REQUIRE(c.n_fft % c.hop_size == 0 && c.n_fft / c.hop_size <= 8 && c.win_size == c.n_fft && c.hop_size % 32 == 0,
                 QTTS_ERR_ARG, "speaker mel: n_fft must be a multiple (<= 8x) of hop_size, win_size == n_fft, hop % 32");
    QTTS_REQUIRE(c.num_mels == c.mel_dim, QTTS_ERR_ARG, "speaker: num_mels must equal the encoder's mel_dim");
    QTTS_REQUIRE(c.res2net_scale >= 2 && c.channels[1] % c.res2net_scale == 0, QTTS_ERR_ARG, "speaker: res2net scale");
    // ---- front end: Hann-windowed DFT as a tap GEMM, mel filterbank (bound by the caller: "mel_basis" (num_mels, n_fft/2+1))
    nb = c.n_fft / 2 + 1; Kp = ((nb + 31) / 32) * 32; ntap = c.n_fft / c.hop_size;
    {
        const int hop = c.hop_size, N = 2 * nb;
        std::vector<float> w((size_t)ntap * N * hop);
        const double two_pi = 6.283185307179586476925286766559;
        for (int tap = 0; tap < ntap; ++tap)
            for (int f = 0; f < nb; ++f)
                for (int j = 0; j < hop; ++j) {
                    const int n = tap * hop + j;
                    const double hann = 0.5 - 0.5 * cos(two_pi * n / c.n_fft);          // torch.hann_window (periodic)
                    const double ang = two_pi * (double)f * (double)n / (double)c.n_fft;
                    w[((size_t)tap * N + f) * hop + j] = (float)(hann * cos(ang));
                    w[((size_t)tap * N + nb + f) * hop + j] = (float)(-hann * sin(ang));
                }
        dft.f32 = true; dft.N = N; dft.K = hop; dft.taps = ntap;
        for (int tap = 0; tap < ntap; ++tap) dft.shift[tap] = -(ntap - 1 - tap);
        upload_w(dft, w);
        auto& fb = P("mel_basis", {c.num_mels, nb});
        std::vector<float> m((size_t)c.num_mels * Kp, 0.f);
        for (int i = 0; i < c.num_mels; ++i)
            for (int f = 0; f < nb; ++f) m[(size_t)i * Kp + f] = fb[(size_t)i * nb + f];
        melw.f32 = true; melw.N = c.num_mels; melw.K = Kp; melw.taps = 1;
        upload_w(melw, m);
    }
    // ---- ECAPA-TDNN
    const int nblk = c.n_blocks;
    make_conv(tdnn0, "blocks.0.conv", c.channels[0], c.mel_dim, c.kernel_sizes[0], c.dilations[0]);
    blks.resize(nblk - 2);
    for (int i = 1; i < nblk - 1; ++i) {
        const std::string p = "blocks." + std::to_string(i) + ".";
        auto& b = blks[i - 1];
        const int ci = c.channels[i - 1], co = c.channels[i], cw = co / c.res2net_scale;
        QTTS_REQUIRE(ci == co, QTTS_ERR_ARG, "speaker: SE-Res2Net blocks need equal in/out channels (residual add)");
        make_conv(b.tdnn1, p + "tdnn1.conv", co, ci, 1, 1);
        b.res.resize(c.res2net_scale - 1);
        for (int j = 0; j < c.res2net_scale - 1; ++j)
            make_conv(b.res[j], p + "res2net_block.blocks." + std::to_string(j) + ".conv", cw, cw, c.kernel_sizes[i], c.dilations[i]);
        make_conv(b.tdnn2, p + "tdnn2.conv", co, co, 1, 1);
        make_conv(b.se1, p + "se_block.conv1", c.se_channels, co, 1, 1);
        make_conv(b.se2, p + "se_block.conv2", co, c.se_channels, 1, 1);
        b.dil = c.dilations[i];
    }
    const int cl = c.channels[nblk - 1];
    int cat = 0;
    for (int i = 1; i < nblk - 1; ++i) cat += c.channels[i];
    QTTS_REQUIRE(cat == cl, QTTS_ERR_ARG, "speaker: mfa expects sum(enc_channels[1:-1]) == enc_channels[-1]");
    make_conv(mfa, "mfa.conv", cl, cl, c.kernel_sizes[nblk - 1], c.dilations[nblk - 1]);
    make_conv(asp_tdnn, "asp.tdnn.conv", c.attention_channels, cl * 3, 1, 1);
    make_conv(asp_conv, "asp.conv", cl, c.attention_channels, 1, 1);
    make_conv(fc, "fc", c.enc_dim, cl * 2, 1, 1);
    // ---- workspace: rows x channels of the widest stage, per sequence
    const int pad = (c.n_fft - c.hop_size) / 2;
    const int64_t R = ((int64_t)c.max_samples + 2 * pad) / c.hop_size + 1;
    const int64_t T = R;                                              // mel frames <= rows
    size_t per_seq = (size_t)R * std::max(2 * nb, Kp);
    per_seq = std::max(per_seq, (size_t)(T + 64) * (size_t)(3 * cl));
    buf_elems = per_seq * (size_t)std::max(1, c.max_batch);
    for (auto& b : buf) b.alloc(buf_elems * sizeof(float));
    host.clear();
    finalized = true;
}

void qtts_speaker::embed(const float* wav, int B, int S, float* out, float* mels_out, hipStream_t st) {
    const auto& c = cfg;
    QTTS_REQUIRE(finalized, QTTS_ERR_STATE, "speaker: finalize() first");
    QTTS_REQUIRE(B >= 1 && B <= c.max_batch && S >= 1 && S <= c.max_samples, QTTS_ERR_LIMIT,
                 "speaker: batch / samples exceed max_batch / max_samples given at create");
    float* w0 = buf[0].as<float>(); float* w1 = buf[1].as<float>(); float* w2 = buf[2].as<float>();
    float* w3 = buf[3].as<float>(); float* w4 = buf[4].as<float>(); float* w5 = buf[5].as<float>();
    // ---- log-mel spectrogram (M:402-464, center = False): reflect pad, frames = rows t .. t+ntap-1
    const int hop = c.hop_size, pad = (c.n_fft - hop) / 2;
    const int R = (S + 2 * pad) / hop;
    const int T = R - (ntap - 1);
    QTTS_REQUIRE(T >= 1, QTTS_ERR_ARG, "speaker: waveform shorter than one STFT window");
    launch_reflect_rows_1d(wav, S, pad, R, hop, w0, B, st);                                   // w0: [B*R][hop]
    gemm(dft, w0, hop, B * R, R, w1, 2 * nb, st);                                             // w1: [B*R][2*nb] (re | im)
    launch_magnitude_pad(w1, 2 * nb, nb, w0, Kp, (int64_t)B * R, st);                         // w0: [B*R][Kp]
    gemm(melw, w0, Kp, B * R, R, w1, c.num_mels, st);                                         // w1: [B*R][num_mels]
    float* mels = w2;                                                                         // [B*T][mel_dim]
    launch_copy_act_rows(w1, c.num_mels, R, ntap - 1, T, c.num_mels, ROWACT_LOG_CLAMP, mels, c.mel_dim, B, st);
    if (mels_out)
        QTTS_CHECK_HIP(hipMemcpyAsync(mels_out, mels, (size_t)B * T * c.mel_dim * sizeof(float), hipMemcpyDeviceToDevice, st));
    // ---- ECAPA-TDNN (M:369-393)
    const int nblk = c.n_blocks, C0 = c.channels[0], cl = c.channels[nblk - 1];
    float* h0 = w3;                                                                           // [B*T][C0]
    tdnn(tdnn0, mels, c.mel_dim, nullptr, 0, B, T, ROWACT_RELU, h0, C0, w0, w1, st);
    float* cat = w4;                                                                          // [B*T][cl]: the block outputs side by side
    int off = 0;
    const float* in = h0; int ld_in = C0;
    for (size_t i = 0; i < blks.size(); ++i) {
        auto& b = blks[i];
        const int co = b.tdnn1.N, cw = co / c.res2net_scale;
        float* h1 = w2;                                                                       // mels are no longer needed
        tdnn(b.tdnn1, in, ld_in, nullptr, 0, B, T, ROWACT_RELU, h1, co, w0, w1, st);
        float* h2 = w5;                                                                       // Res2Net output [B*T][co]
        launch_copy_act_rows(h1, co, T, 0, T, cw, ROWACT_NONE, h2, co, B, st);                // group 0 passes through
        for (int s = 1; s < c.res2net_scale; ++s)
            tdnn(b.res[s - 1], h1 + s * cw, co, s >= 2 ? h2 + (s - 1) * cw : nullptr, co, B, T, ROWACT_RELU, h2 + s * cw, co, w0, w1, st);
        float* h3 = w2;                                                                       // h1 is free again
        tdnn(b.tdnn2, h2, co, nullptr, 0, B, T, ROWACT_RELU, h3, co, w0, w1, st);
        // squeeze-excitation (M:129-158): gate = sigmoid(conv2(relu(conv1(mean_t h3))))
        float* mean = w0; float* g1 = w0 + (size_t)B * co; float* tmp = w1; float* gate = w5;  // h2 is free again
        launch_col_stats(h3, co, nullptr, T, co, mean, nullptr, co, B, st);
        gemm(b.se1, mean, co, B, 1, tmp, c.se_channels, st);
        launch_copy_act_rows(tmp, c.se_channels, 1, 0, 1, c.se_channels, ROWACT_RELU, g1, c.se_channels, B, st);
        gemm(b.se2, g1, c.se_channels, B, 1, tmp, co, st);
        launch_copy_act_rows(tmp, co, 1, 0, 1, co, ROWACT_SIGMOID, gate, co, B, st);
        launch_scale_add_rows(h3, co, gate, in, ld_in, cat + off, cl, T, co, B, st);          // + residual, into its slice
        in = cat + off; ld_in = cl; off += co;
    }
    // ---- multi-layer feature aggregation + attentive statistics pooling (M:161-251)
    float* x = w3;                                                                            // [B*T][cl]  (h0 is free)
    tdnn(mfa, cat, cl, nullptr, 0, B, T, ROWACT_RELU, x, cl, w0, w1, st);
    float* mean = w2; float* sd = w2 + (size_t)B * cl;
    launch_col_stats(x, cl, nullptr, T, cl, mean, sd, cl, B, st);
    float* ain = w4;                                                                          // [B*T][3*cl]  (cat is free)
    launch_concat_stats(x, cl, mean, sd, T, cl, ain, B, st);
    float* a1 = w5;                                                                           // [B*T][attention_channels]
    tdnn(asp_tdnn, ain, 3 * cl, nullptr, 0, B, T, ROWACT_RELU_TANH, a1, c.attention_channels, w0, w1, st);
    float* a2 = w4;                                                                           // [B*T][cl]
    gemm(asp_conv, a1, c.attention_channels, B * T, T, a2, cl, st);
    launch_softmax_time(a2, T, cl, B, st);
    float* pooled = w2;                                                                       // [B][2*cl] = mean | std
    launch_col_stats(x, cl, a2, T, cl, pooled, pooled + cl, 2 * cl, B, st);
    gemm(fc, pooled, 2 * cl, B, 1, out, c.enc_dim, st);
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

int qtts_speaker_create(const qtts_speaker_config* cfg, qtts_speaker** out) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(cfg && out, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(cfg->compute_dtype == QTTS_F32 || cfg->compute_dtype == QTTS_BF16, QTTS_ERR_ARG, "compute_dtype");
    int ndev = 0;
    if (!getenv("QTTS_DEBUG_NO_DEVICE")) {
        QTTS_CHECK_HIP(hipGetDeviceCount(&ndev));
        QTTS_REQUIRE(ndev > 0, QTTS_ERR_HIP, "no HIP device");
    }
    auto* s = new qtts_speaker();
    s->cfg = *cfg;
    s->bf16 = cfg->compute_dtype == QTTS_BF16;
    *out = s;
    QTTS_API_END
}
void qtts_speaker_destroy(qtts_speaker* s) { delete s; }
int qtts_speaker_bind(qtts_speaker* s, const char* name, const void* hostp, int32_t src_dtype, int32_t ndim, const int64_t* shape) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(s && name && hostp && shape, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(!s->finalized, QTTS_ERR_STATE, "bind after finalize");
    QTTS_REQUIRE(src_dtype == QTTS_F32 || src_dtype == QTTS_BF16, QTTS_ERR_ARG, "src_dtype");
    HostTensor ht{hostp, src_dtype, std::vector<int64_t>(shape, shape + ndim)};
    s->host[name] = ht.to_f32();
    s->shapes[name] = ht.shape;
    QTTS_API_END
}
int qtts_speaker_finalize(qtts_speaker* s) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(s, QTTS_ERR_ARG, "null handle");
    s->finalize();
    QTTS_API_END
}
int qtts_speaker_mel_frames(qtts_speaker* s, int64_t samples, int64_t* frames) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(s && frames && samples >= 1, QTTS_ERR_ARG, "bad argument");
    QTTS_REQUIRE(s->finalized, QTTS_ERR_STATE, "speaker: finalize() first");
    *frames = s->mel_frames(samples);
    QTTS_API_END
}
int qtts_speaker_embed(qtts_speaker* s, const float* wav_dev, int32_t B, int32_t samples, float* emb_dev, float* mels_dev,
                       void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(s && wav_dev && emb_dev, QTTS_ERR_ARG, "null argument");
    s->embed(wav_dev, B, samples, emb_dev, mels_dev, (hipStream_t)stream);
    QTTS_API_END
}

}  // extern "C"
