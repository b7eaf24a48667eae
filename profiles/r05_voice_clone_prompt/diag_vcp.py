import sys, time, os, gc
sys.path.insert(0, os.getcwd())      # (run from the repository root)
import numpy as np, torch, synth
from qwen3_tts_amd.codec import Qwen3TTSTokenizer
from qwen3_tts_amd.speaker import SpeakerEncoderEngine
dev = torch.device("cuda:0")
td = lambda w: {k: torch.from_numpy(v) for k, v in w.items()}
enc, spk, dec = synth.mimi_enc_real(), synth.speaker_real(2048), synth.codec_real()
n = 72000
tok_sd = dict(synth.codec_weights(dec)); tok_sd.update({"encoder." + k: v for k, v in synth.mimi_enc_weights(enc).items()})
tok_cfg = dict(synth.cfg_dict(dec), encoder_config=synth.cfg_dict(enc), encoder_valid_num_quantizers=enc.encoder_valid_num_quantizers,
               encode_downsample_rate=enc.encode_downsample_rate, input_sample_rate=24000)
dt = torch.bfloat16
tok = Qwen3TTSTokenizer.from_state_dict(tok_cfg, td(tok_sd), device=dev, dtype=dt, max_batch=32, max_frames=32)
se = SpeakerEncoderEngine(synth.cfg_dict(spk), td(synth.speaker_weights(spk)), compute_dtype=dt, device=dev, max_batch=1, max_samples=n)
clips = [a for a in synth.rand_audio(308, 8, n)]
tok.encode(clips, sr=24000); torch.cuda.synchronize()
E = tok.model._encoder
x_host = torch.from_numpy(np.stack(clips))
x_dev = x_host.to(dev)
for mode in ("full", "nogc", "dev_input"):
    if mode == "nogc": gc.disable()
    ts = []
    for rep in range(16):
        ta = time.perf_counter()
        if mode in ("full", "nogc"): tok.encode(clips, sr=24000)
        elif mode == "dev_input": E.encode_padded(x_dev)
        else: E.encode_padded(x_dev)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - ta))
    print("encode", mode, " ".join("%.1f" % t for t in ts), flush=True)
for mode in ("host", "dev"):
    ts = []
    for rep in range(16):
        ta = time.perf_counter()
        if mode == "host":
            for a in clips: se.extract_speaker_embedding(a, 24000)
        else:
            for i in range(8): se.embed(x_dev[i:i + 1])
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - ta))
    print("speaker", mode, " ".join("%.1f" % t for t in ts), flush=True)
