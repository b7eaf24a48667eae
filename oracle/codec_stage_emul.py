"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Python mirror of `qtts_codec::stream_push` (qwen3-tts_amd/csrc/codec_engine.hip): the same staging / skip / carry
bookkeeping, expressed with the oracle's whole-buffer ops -- zero-left-padded causal convs and left-pad-masked
sliding-window attention run on [carried rows | new rows] buffers, the outputs of the carried rows dropped at the next
staging step.  It checks the ALGEBRA the C++ orchestration relies on (carry sizes and order, how `skip` scales through
the transposed convs, the n_pad of the staged KV window) against the whole-sequence forward on the CPU
(tests/test_oracle_golden.py); the kernels it stands in for are the validated ones of the non-streaming path."""
import torch
import torch.nn.functional as F

import codec_ref as R
from codec_ref import _t

class StagedStream:
    def __init__(self, w, c, B):
        self.w, self.c, self.B = w, c, B
        self.carry = None
        self.t = 0
    def begin(self):
        c = self.c
        want = [(2, c.codebook_dim)]
        qkvw = (c.num_attention_heads + 2 * c.num_key_value_heads) * c.head_dim
        want += [(c.sliding_window - 1, qkvw)] * c.num_hidden_layers
        want += [(6, c.latent_dim)] * len(c.upsampling_ratios)
        want += [(6, c.latent_dim)]
        for i in range(len(c.upsample_rates)):
            want.append((1, c.decoder_dim >> i))
            for d in (1, 3, 9):
                want.append((6 * d, c.decoder_dim >> (i + 1)))
        want.append((6, c.decoder_dim >> len(c.upsample_rates)))
        self.carry = [torch.zeros(self.B, C, h) for h, C in want]     # channel-first here
        self.t = 0
    def push(self, codes):
        w, c, B = self.w, self.c, self.B
        n = codes.shape[-1]
        ci = [0]
        st = {"x": None, "T": n, "skip": 0}
        def stage():
            k = self.carry[ci[0]]
            x = st["x"]
            assert k.shape[1] == x.shape[1], (ci[0], k.shape, x.shape)
            d = torch.cat([k, x[..., st["skip"]:]], dim=-1)
            h = k.shape[-1]
            self.carry[ci[0]] = d[..., d.shape[-1] - h:].clone() if h > 0 else k
            ci[0] += 1
            st["x"], st["T"], st["skip"] = d, d.shape[-1], h
        def compact():
            st["x"] = st["x"][..., st["skip"]:]; st["T"] = st["x"].shape[-1]; st["skip"] = 0
        st["x"] = R.rvq_dequant(w, c, codes)
        stage()
        st["x"] = R.causal_conv1d(st["x"], _t(w, "pre_conv.conv.weight"), _t(w, "pre_conv.conv.bias"))
        compact()
        # transformer
        x = st["x"].transpose(1, 2)                       # (B, n, Ld)
        nh, nkv, hd, W = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.sliding_window
        W1 = W - 1
        qd, kvd = nh * hd, nkv * hd
        p = "pre_transformer."
        h = F.linear(x, _t(w, p + "input_proj.weight"), _t(w, p + "input_proj.bias"))
        pad = W1 - min(self.t, W1)
        cos, sin = R._rope_cos_sin(torch.arange(self.t, self.t + n), hd, c.rope_theta)
        for l in range(c.num_hidden_layers):
            lp = f"{p}layers.{l}."
            a = R._rmsnorm(h, _t(w, lp + "input_layernorm.weight"), c.rms_norm_eps)
            q = F.linear(a, _t(w, lp + "self_attn.q_proj.weight")).view(B, n, nh, hd)
            k = F.linear(a, _t(w, lp + "self_attn.k_proj.weight")).view(B, n, nkv, hd)
            v = F.linear(a, _t(w, lp + "self_attn.v_proj.weight"))
            q = (q.transpose(1, 2) * cos + R._rotate_half(q.transpose(1, 2)) * sin).transpose(1, 2).reshape(B, n, qd)
            k = (k.transpose(1, 2) * cos + R._rotate_half(k.transpose(1, 2)) * sin).transpose(1, 2).reshape(B, n, kvd)
            qkv = torch.cat([q, k, v], dim=-1).transpose(1, 2)            # channel-first (B, qkvw, n)
            kc = self.carry[ci[0]]
            S = torch.cat([kc, qkv], dim=-1)                               # (B, qkvw, W1+n)
            self.carry[ci[0]] = S[..., S.shape[-1] - W1:].clone()
            ci[0] += 1
            Tp = W1 + n
            Sq = S.transpose(1, 2)
            qq = Sq[..., :qd].reshape(B, Tp, nh, hd).transpose(1, 2)
            kk = Sq[..., qd:qd + kvd].reshape(B, Tp, nkv, hd).transpose(1, 2)
            vv = Sq[..., qd + kvd:].reshape(B, Tp, nkv, hd).transpose(1, 2)
            if nkv != nh:
                kk = kk.repeat_interleave(nh // nkv, 1); vv = vv.repeat_interleave(nh // nkv, 1)
            qi = torch.arange(Tp)[:, None]; ki = torch.arange(Tp)[None, :]
            allowed = (ki <= qi) & (ki > qi - W) & (ki >= pad)
            bias = torch.zeros(Tp, Tp).masked_fill(~allowed, float("-inf"))
            att = torch.matmul(qq, kk.transpose(2, 3)) * hd ** -0.5 + bias
            att = torch.softmax(att, dim=-1)
            att = torch.nan_to_num(att)                                    # fully masked (skipped) rows
            o = torch.matmul(att, vv).transpose(1, 2).reshape(B, Tp, qd)[:, W1:]
            h = h + _t(w, lp + "self_attn_layer_scale.scale") * F.linear(o, _t(w, lp + "self_attn.o_proj.weight"))
            a = R._rmsnorm(h, _t(w, lp + "post_attention_layernorm.weight"), c.rms_norm_eps)
            m = F.linear(F.silu(F.linear(a, _t(w, lp + "mlp.gate_proj.weight"))) * F.linear(a, _t(w, lp + "mlp.up_proj.weight")),
                         _t(w, lp + "mlp.down_proj.weight"))
            h = h + _t(w, lp + "mlp_layer_scale.scale") * m
        h = R._rmsnorm(h, _t(w, p + "norm.weight"), c.rms_norm_eps)
        st["x"] = F.linear(h, _t(w, p + "output_proj.weight"), _t(w, p + "output_proj.bias")).permute(0, 2, 1)
        st["T"], st["skip"] = n, 0
        for u, f in enumerate(c.upsampling_ratios):
            st["x"] = R.causal_transconv1d(st["x"], _t(w, f"upsample.{u}.0.conv.weight"), _t(w, f"upsample.{u}.0.conv.bias"), f)
            st["T"] *= f; st["skip"] *= f
            stage()
            st["x"] = R.convnext(w, f"upsample.{u}.1.", st["x"])
        stage()
        st["x"] = R.causal_conv1d(st["x"], _t(w, "decoder.0.conv.weight"), _t(w, "decoder.0.conv.bias"))
        for i, r in enumerate(c.upsample_rates):
            pb = f"decoder.{i + 1}.block."
            stage()
            a = R.snake_beta(st["x"], _t(w, pb + "0.alpha"), _t(w, pb + "0.beta"))
            st["x"] = R.causal_transconv1d(a, _t(w, pb + "1.conv.weight"), _t(w, pb + "1.conv.bias"), r)
            st["T"] *= r; st["skip"] *= r
            for j, d in zip((2, 3, 4), (1, 3, 9)):
                stage()
                st["x"] = R.res_unit(w, pb + f"{j}.", st["x"], d)
        nn_ = len(c.upsample_rates)
        stage()
        a = R.snake_beta(st["x"], _t(w, f"decoder.{nn_ + 1}.alpha"), _t(w, f"decoder.{nn_ + 1}.beta"))
        y = R.causal_conv1d(a, _t(w, f"decoder.{nn_ + 2}.conv.weight"), _t(w, f"decoder.{nn_ + 2}.conv.bias"))
        assert ci[0] == len(self.carry)
        self.t += n
        return y[..., st["skip"]:].clamp(-1, 1)
