#!/usr/bin/env python
"""tools/bench_configs.py -- the two BASELINE.json configurations that are NOT the bench.py line:

  first_packet   config 4: Qwen3-TTS-12Hz-1.7B dims, VoiceDesign-shaped batch of 32, streaming text input (text fed one
                 token per frame through `trailing_text_hidden`, M:2229-2232).  The reference has no streaming OUTPUT
                 API (IM:513-515), so first packet = wall time from the call to the first 4 frames (320 ms of audio)
                 decoded to PCM on the host: prefill + first token + 4 frame steps + codec decode of frames [0,4).
                 Reports p50 / p99 over trials (all 32 rows of a batch finish together) and the per-leg split.
  clone_shard    config 5: 256 voice-clone (ICL) requests -- 38 reference frames + ref text in the prompt -- dealt to
                 the ranks longest-first (sharding.lpt_partition), each rank running waves of 8, waveforms gathered on
                 rank 0 with ONE torch.distributed.gather (nccl = RCCL).  Launch with torch.distributed.run for N > 1.

Synthetic seeded weights and prompts; prints one JSON line on rank 0.  Not the judged bench line -- that is bench.py."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _engines(args, B, max_seq, max_frames, dev):
    import torch
    import synth
    from qwen3_tts_amd.codec import CodecDecoderEngine
    from qwen3_tts_amd.talker import TalkerEngine
    tcfg = {"1.7b": synth.talker_17b, "0.6b": synth.talker_06b, "tiny": synth.talker_tiny}[args.model]()
    ccfg = synth.codec_tiny() if args.model == "tiny" else synth.codec_real()
    td = lambda w: {k: torch.from_numpy(v) for k, v in w.items()}
    talker = TalkerEngine(tcfg, td(synth.talker_weights(tcfg, with_text=False)), weight_dtype=torch.bfloat16, device=dev,
                          max_batch=B, max_seq=max_seq, use_graph=True)
    codec = CodecDecoderEngine(ccfg, td(synth.codec_weights(ccfg)), compute_dtype=torch.bfloat16, device=dev, max_batch=B,
                               max_frames=max_frames)
    return tcfg, ccfg, talker, codec


def _sampling(tcfg):
    sup = [i for i in range(tcfg.vocab_size - 1024, tcfg.vocab_size) if i != tcfg.codec_eos_token_id]
    return dict(suppress_tokens=sup, repetition_penalty=1.05, output_hidden_states=False, do_sample=True, top_k=50, top_p=1.0,
                temperature=0.9, subtalker_dosample=True, subtalker_top_k=50, subtalker_top_p=1.0, subtalker_temperature=0.9)


def first_packet(args):
    import numpy as np
    import torch
    import synth
    dev = "cuda:0"
    B, NF = args.batch, 4
    text = [24 + 4 * (i % 8) for i in range(B)]                # text tokens, fed one per frame
    lens = [32 + 12 + (i % 5) for i in range(B)]               # instruct (32) + role / codec prefix rows, ragged
    tcfg, ccfg, talker, codec = _engines(args, B, max(lens) + 16, 32, dev)
    rng = np.random.default_rng(4)
    emb, mask, trailing, pad = [x.to(dev) for x in synth.rand_prompt(rng, tcfg, lens, max(text), 0.05)]
    kw = dict(_sampling(tcfg), max_new_tokens=NF + 1, min_new_tokens=NF + 1)
    lat, legs = [], []
    for trial in range(args.trials + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = talker.generate(emb, mask, trailing, pad, seed=50 + trial, **kw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        wav, wl = codec.decode_padded(out.codes[:, :NF])
        pcm = wav.cpu()                                        # first packet is on the host here
        t2 = time.perf_counter()
        assert out.n_frames == NF and pcm.shape[1] == NF * ccfg.total_upsample
        if trial >= 2:
            lat.append(1e3 * (t2 - t0))
            legs.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    lat = np.array(lat)
    return {"config": "first_packet", "model": args.model, "batch": B, "frames_in_packet": NF, "audio_ms_in_packet": NF * 80,
            "trials": len(lat), "p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
            "min_ms": round(float(lat.min()), 3), "prefill_plus_ar_ms_p50": round(float(np.median([a for a, _ in legs])), 3),
            "codec_plus_d2h_ms_p50": round(float(np.median([b for _, b in legs])), 3),
            "note": "wall time from generate() call to 4 frames of PCM on the host; streaming text input; sampling"}


def clone_shard(args):
    import numpy as np
    import torch
    import synth
    from qwen3_tts_amd import sharding
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = f"cuda:{local_rank}"
    NREQ, B, REF = args.requests, 8, 38
    rq = np.random.default_rng(5)
    text = rq.integers(16, 72, NREQ).tolist()                   # text tokens per request
    frames = [int(round(2.2 * t)) for t in text]                # synthetic length model: frames ~ text tokens
    parts = sharding.lpt_partition(text, world)                 # every rank computes the same partition
    mine = parts[rank]
    my_waves = sharding.waves(sorted(mine, key=lambda i: -text[i]), B)   # similar lengths share a wave
    Fmax = max(frames)
    # --engines E: E independent (talker, codec) engine pairs on this GPU, each on its own HIP stream and host thread
    # (ctypes releases the GIL inside the C calls).  The frame step is latency-bound -- ~570 dependent launches that each
    # leave most of the 256 CUs idle -- so a second stream can fill the gaps; weights are replicated per engine (3.9 GB each).
    E = max(1, args.engines)
    pairs = [_engines(args, B, REF + 16 + 72 + 12 + Fmax + 8, min(Fmax, 300) + 25, dev) for _ in range(E)]
    tcfg, ccfg = pairs[0][0], pairs[0][1]
    base = _sampling(tcfg)

    def run_wave(w, seed, e=0):
        talker, codec = pairs[e][2], pairs[e][3]
        # ICL prompt (M:1968-2019): role + codec prefix + [ref text + text] over [ref codes] -> lens = 12 + REF + ref text (16)
        lens = [12 + REF + 16 + (text[i] % 7) for i in w]
        g = np.random.default_rng(1000 + w[0])
        emb, mask, trailing, pad = [x.to(dev) for x in synth.rand_prompt(g, tcfg, lens, max(text[i] for i in w), 0.05)]
        F = max(frames[i] for i in w)
        out = talker.generate(emb, mask, trailing, pad, seed=seed, max_new_tokens=F + 1, min_new_tokens=F + 1, **base)
        codes = out.codes.clone()
        for j, i in enumerate(w):                               # each request keeps its own length (rest = -1 padding)
            codes[j, frames[i]:] = -1
        wav, wl = codec.decode_padded(codes)
        return [wav[j, :int(wl[j])] for j in range(len(w))]

    for e in range(E):
        run_wave(my_waves[0], 1, e)                              # warm-up (graph capture, allocator) on every engine
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    local_wavs, local_idx = [], []
    if E == 1:
        for k, w in enumerate(my_waves):
            for i, x in zip(w, run_wave(w, 100 + k)):
                local_idx.append(i)
                local_wavs.append(x)
    else:
        import threading
        results, errors = [None] * len(my_waves), []

        def worker(e):
            try:
                torch.cuda.set_device(local_rank)
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):   # per-thread current stream: copies and the codec
                    for k in range(e, len(my_waves), E):                 # stay off the shared default stream
                        results[k] = run_wave(my_waves[k], 100 + k, e)   # static round-robin: wave k -> engine k % E
                    torch.cuda.current_stream().synchronize()
            except Exception as ex:                              # surface worker failures on the main thread
                errors.append(ex)

        th = [threading.Thread(target=worker, args=(e,)) for e in range(E)]
        [t.start() for t in th]
        [t.join() for t in th]
        if errors:
            raise errors[0]
        for w, r in zip(my_waves, results):
            for i, x in zip(w, r):
                local_idx.append(i)
                local_wavs.append(x)
    torch.cuda.synchronize()
    t_compute = time.perf_counter() - t0
    if dist is not None:
        allw = sharding.gather_waveforms([x.cpu().numpy() for x in local_wavs], local_idx, NREQ)
        dist.barrier()
    else:
        allw = [None] * NREQ
        for i, x in zip(local_idx, local_wavs):
            allw[i] = x.cpu().numpy()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed, t_compute], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, t_compute = float(tt[0]), float(tt[1])
    if rank != 0:
        return None
    assert all(allw[i] is not None and allw[i].shape[0] == frames[i] * ccfg.total_upsample for i in range(NREQ))
    tok = sum(frames) * tcfg.num_code_groups
    audio_s = sum(frames) * ccfg.total_upsample / 24000.0
    return {"config": "clone_shard", "model": args.model, "requests": NREQ, "n_gpus": world, "engines_per_gpu": E,
            "waves_rank0": len(my_waves),
            "seconds": round(elapsed, 3), "compute_seconds_max_rank": round(t_compute, 3),
            "speech_tokens_per_s": round(tok / elapsed, 1), "rtf_x": round(audio_s / elapsed, 2),
            "padding_waste": round(1.0 - sum(frames) / sum(max(frames[i] for i in w) * len(w) for p in parts
                                                         for w in sharding.waves(sorted(p, key=lambda i: -text[i]), B)), 3),
            "rank_load_imbalance": round(max(sum(text[i] for i in p) for p in parts) / (sum(text) / world), 3)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["first_packet", "clone_shard"])
    ap.add_argument("--model", default="1.7b", choices=["1.7b", "0.6b", "tiny"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--trials", type=int, default=20)
    ap.add_argument("--requests", type=int, default=256)
    ap.add_argument("--engines", type=int, default=1, help="clone_shard: engine pairs per GPU running waves concurrently")
    a = ap.parse_args()
    r = first_packet(a) if a.config == "first_packet" else clone_shard(a)
    if r is not None:
        print(json.dumps(r))
