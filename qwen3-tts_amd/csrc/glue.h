// glue.h -- device-resident loop state and the small frame-step kernels (sampling.hip).
#pragma once
#include "common.h"

namespace qtts {

// Everything that varies per decode step lives in device memory so the captured frame step is static.
struct StepState {
    int* n_generated;   // tokens sampled so far (HF cur_len of the generated part)
    int* gen_step;      // talker `generation_step` of the NEXT decode forward (= frame index)
    int* kv_len;        // talker KV length before the next decode forward
    int* done;          // latched when HF's stopping criteria would break
    int* final_count;   // n_generated at that moment
    int* unfinished;    // [B]
};
void launch_sample_finish(const StepState& s, int B, int max_new_tokens, hipStream_t st);

struct CpGatherParams {
    int pass, B, H;
    const float* past_hidden;   // [B][H]
    const float* talker_emb;    // talker codec_embedding [vocab][H]
    const int* cur_tok;         // [B]
    const float* cp_emb;        // code-predictor codec_embedding [G-1][cp_vocab][H]
    int cp_vocab;
    const int* sub; int sub_stride;   // [B][G-1] sub-codes sampled so far
    float* out;                 // [rows][H]
    unsigned short* out16;      // optional bf16 copy [rows][H]
    const int* done;
};
void launch_cp_gather(const CpGatherParams& p, hipStream_t st);

struct EmbedSumParams {
    int B, H, G, cp_vocab;
    const float* talker_emb; const float* cp_emb;
    const int* cur_tok; const int* sub; int sub_stride;
    const float* trailing; int Tt; const float* tts_pad;
    const float* past_hidden;
    float* x_out; unsigned short* x_out16;   // fp32 hidden state (+ optional bf16 copy)
    int64_t* codes_out; float* hidden_out; int max_frames;
    StepState st;
};
void launch_embed_sum(const EmbedSumParams& p, hipStream_t st);

// y = g * (x * rsqrt(mean(x^2)+eps)) per row (final talker norm -> past_hidden), early-exit on *done
void launch_apply_norm(const float* x, int ldx, const float* g, float eps, float* y, int ldy, int rows, int C,
                       const int* done, hipStream_t st);
// ss[r] = sum_c x[r][c]^2 (only for GEMMs that cannot stage x through LDS)
void launch_row_ss(const float* x, int ldx, int rows, int C, float* ss, const int* done, hipStream_t st);

}  // namespace qtts
