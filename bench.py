#!/usr/bin/env python
"""bench.py -- the north-star benchmark of the MI355X-native Qwen3-TTS hot path.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic requests, inputs resident in HBM:
talker prefill -> `frames` autoregressive frame steps (15-pass code predictor + 28-layer talker + on-device
sampling, hipGraph) -> Qwen3-TTS-Tokenizer-12Hz codec decode of every utterance to a 24 kHz waveform.
Workload = BASELINE.json's metric config: Qwen3-TTS-12Hz-1.7B dims, batch 8, 10 s (125 frames) per utterance,
default sampling (T 0.9, top-k 50, rep 1.05), seeded random weights (no checkpoint exists offline), bf16.

With N > 1 every rank runs its own batch (request sharding, no data-path collective) and every step ends
with the shard's one exchange -- the waveforms gathered on rank 0 (`sharding.gather_padded`, nccl = RCCL over xGMI), inside
the timed region: weak scaling; value = all ranks' speech tokens / max-over-ranks time.

`python bench.py --gpus N` with N > 1 and no torch.distributed environment launches the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) and fails loudly when the
box has fewer than N GPUs; under an existing launcher (WORLD_SIZE set) it is one rank of that job.

Prints ONE JSON line on rank 0.  Extra legs (rank 0, N = 1 only, outside the timed region):
  roofline      HIP-event timing of every launch of the dominant kernel (skinny weight-streaming GEMM) over one
                eager generate call, against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (oracle/*_ref.py: torch fp32 restatement of the reference) on a bounded sample of
                the same workload, timed on this host's cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
_T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def synth_prompt(rng, cfg, lens, n_trail, scale=0.05):
    import synth
    return synth.rand_prompt(rng, cfg, lens, n_trail, scale)


def self_launch(n_gpus, argv, backend):
    """`--gpus N` (N > 1) outside a torch.distributed launcher: start the N ranks of this node here, one per GPU, and
    return the launcher's exit code.  Refuses (exit code 2) when the box has fewer than N GPUs -- never a silent 1-rank run."""
    import socket
    import subprocess
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n_gpus:
            print(f"[bench] FATAL: --gpus {n_gpus} but this box exposes {have} GPU(s); refusing to run fewer ranks than asked",
                  file=sys.stderr, flush=True)
            return 2
    with socket.socket() as sk:                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0])] + list(argv)
    log("self-launch: " + " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main(argv=None, device=None):
    """`device` is for test wrappers only (tests/hostemu/bench_emu.py runs the launcher path on a CPU container against the
    host-emulation build, which IT installs): any line produced with it is marked INVALID.  bench.py itself never loads
    anything but qwen3-tts_amd/libqtts.so on a HIP device."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="1.7b", choices=["1.7b", "0.6b", "tiny"])
    ap.add_argument("--batch", type=int, default=None, help="requests per engine batch: 8 for the metric workload, 32 (waves) for clone-shard")
    ap.add_argument("--engines", type=int, default=None, help="clone-shard: engine pairs per GPU running waves concurrently (default 2)")
    ap.add_argument("--frames", type=int, default=125, help="codec frames per utterance (125 = 10 s)")
    ap.add_argument("--greedy", action="store_true", help="greedy decode instead of the default sampling")
    ap.add_argument("--codec-dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--talker-dtype", default="bf16", choices=["bf16", "f32"],
                    help="bf16 = the benchmarked mode (the reference examples' dtype); f32 = the exact-fp32 parity mode")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--overlap-codec", type=int, default=0, metavar="FRAMES",
                    help="EXPERIMENT (off by default, not the reported configuration): decode packets of FRAMES frames through the "
                         "state-carrying stream decoder while the talker generates the next ones (streaming-output orchestration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=40, help="frames of the bounded CPU-baseline sample")
    ap.add_argument("--cpu-budget-s", type=float, default=90.0, help="wall-clock cap of the CPU-baseline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 job (nccl = RCCL; gloo only for the CPU launcher test)")
    ap.add_argument("--no-parity-mode", action="store_true",
                    help="skip the fp32 parity-mode timing leg (fp32 talker frame step + fp32 codec decode, N = 1 only)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the configs leg (BASELINE configs 2-5 on the line: codec-only, 0.6B b8, first packet b32, clone-shard at N = 1; N = 1 only)")
    ap.add_argument("--no-api-e2e", action="store_true",
                    help="skip the api_e2e leg (Qwen3TTSModel.generate_custom_voice: text -> host numpy, N = 1 only)")
    ap.add_argument("--workload", default="metric", choices=["metric", "clone-shard"],
                    help="metric = BASELINE.json's metric config (weak scaling: one batch per GPU); clone-shard = BASELINE "
                         "config 5, a FIXED job of --requests voice-clone requests dealt to the ranks (strong scaling)")
    ap.add_argument("--requests", type=int, default=256, help="clone-shard: requests in the job")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.batch is None:
        args.batch = 32 if args.workload == "clone-shard" else 8
    if args.engines is None:
        args.engines = 2 if args.workload == "clone-shard" else 1
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:] if argv is None else argv, args.backend))
    hostemu = device is not None
    if not hostemu and os.environ.get("QTTS_LIBRARY"):
        from qwen3_tts_amd import _lib as _qlib
        if _qlib.library_override():
            print(f"[bench] FATAL: QTTS_LIBRARY={os.environ['QTTS_LIBRARY']} -- the bench line is only ever measured on the product "
                  "library (qwen3-tts_amd/libqtts.so)", file=sys.stderr, flush=True)
            sys.exit(2)

    import numpy as np
    import torch
    import synth
    from qwen3_tts_amd.codec import CodecDecoderEngine
    from qwen3_tts_amd.talker import TalkerEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.gpus != world:
        print(f"[bench] FATAL: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        from qwen3_tts_amd.sharding import gather_padded
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    dev = device if hostemu else f"cuda:{local_rank}"
    if not hostemu:
        torch.cuda.set_device(local_rank)

    tcfg = {"1.7b": synth.talker_17b, "0.6b": synth.talker_06b, "tiny": synth.talker_tiny}[args.model]()
    ccfg = synth.codec_tiny() if args.model == "tiny" else synth.codec_real()
    if args.model == "tiny":
        ccfg.codebook_size = tcfg.cp_vocab_size          # the codec codebooks must cover the talker's code range
    if args.workload == "clone-shard":
        res = clone_shard_job(args, tcfg, ccfg, dev, rank, world, dist)
        if rank == 0:
            res["frame_steps_in_job"] = sum(res.pop("_wave_frames"))
            if hostemu:
                res["INVALID"] = f"run by a test wrapper on device {device!r}: launcher test only, not a measurement"
            print(json.dumps(res), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    B, F = args.batch, args.frames
    t0 = time.time()
    tw_np = synth.talker_weights(tcfg, with_text=False)
    cw_np = synth.codec_weights(ccfg)
    td = lambda w: {k: torch.from_numpy(v) for k, v in w.items()}
    lens = [24 + 4 * (i % 8) + 12 for i in range(B)]          # 128-char prompts ~ 24..52 text tokens + 12 prefix rows
    talker = TalkerEngine(tcfg, td(tw_np), weight_dtype=torch.bfloat16 if args.talker_dtype == "bf16" else torch.float32, device=dev, max_batch=B,
                          max_seq=max(lens) + F + 8, use_graph=not args.no_graph)
    codec = CodecDecoderEngine(ccfg, td(cw_np), compute_dtype=torch.bfloat16 if args.codec_dtype == "bf16" else torch.float32,
                               device=dev, max_batch=B, max_frames=min(F, 300) + 25)
    build_s = time.time() - t0
    log(f"engines built in {build_s:.1f}s")
    rng = np.random.default_rng(100 + rank)
    emb, mask, trailing, pad = synth_prompt(rng, tcfg, lens, 1)
    emb, mask, trailing, pad = emb.to(dev), mask.to(dev), trailing.to(dev), pad.to(dev)
    sup = [i for i in range(tcfg.vocab_size - 1024, tcfg.vocab_size) if i != tcfg.codec_eos_token_id]
    gen_kw = dict(max_new_tokens=F + 1, min_new_tokens=F + 1, suppress_tokens=sup, repetition_penalty=1.05,
                  output_hidden_states=False)
    if args.greedy:
        gen_kw.update(do_sample=False, subtalker_dosample=False)
    else:
        gen_kw.update(do_sample=True, top_k=50, top_p=1.0, temperature=0.9, subtalker_dosample=True, subtalker_top_k=50,
                      subtalker_top_p=1.0, subtalker_temperature=0.9)

    def step_overlapped(seed):
        # streaming-output orchestration: the talker (own stream) yields packets of codes, each packet goes through the
        # state-carrying codec stream (concatenated packets == whole-sequence decode) while the next frames are generated
        codec.stream_begin(B)
        parts, n = [], 0
        for pk in talker.generate_stream(emb, mask, trailing, pad, packet_frames=args.overlap_codec, seed=seed, **gen_kw):
            parts.append(codec.stream_push(pk.transpose(1, 2).contiguous())[:, 0])
            n += pk.shape[1]
        assert n == F, f"expected {F} frames, got {n}"
        wav = torch.cat(parts, dim=1)
        return None, wav, [wav.shape[1]] * B

    def step(seed):
        if args.overlap_codec > 0:
            return step_overlapped(seed)
        out = talker.generate(emb, mask, trailing, pad, seed=seed, **gen_kw)
        assert out.n_frames == F, f"expected {F} frames, got {out.n_frames}"
        wav, wl = codec.decode_padded(out.codes)
        if dist is not None:                  # the request shard's one exchange: every rank's waveforms land on rank 0
            gather_padded(wav, torch.as_tensor(wl), None)
        return out, wav, wl

    for i in range(args.warmup):
        step(1000 + i)
    log("warmup done")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_ar = 0.0
    t_codec = 0.0
    t_gather = 0.0
    gathered = None
    t1 = time.perf_counter()
    for i in range(args.steps):
        ta = time.perf_counter()
        if args.overlap_codec > 0:          # (no per-leg split in this mode: the legs overlap)
            out, wav, wl = step_overlapped(2000 + i)
            torch.cuda.synchronize()
            t_ar += time.perf_counter() - ta
            continue
        out = talker.generate(emb, mask, trailing, pad, seed=2000 + i, **gen_kw)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        t_ar += tb - ta
        assert out.n_frames == F
        wav, wl = codec.decode_padded(out.codes)
        torch.cuda.synchronize()            # split the two legs for the breakdown fields (a few us per step)
        tc = time.perf_counter()
        t_codec += tc - tb
        if dist is not None:
            gathered = gather_padded(wav, torch.as_tensor(wl), None)
            torch.cuda.synchronize()
            t_gather += time.perf_counter() - tc
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    log(f"timed region: {elapsed:.3f}s for {args.steps} steps")
    per_rank = None
    if dist is not None:
        # what every rank saw, collected THROUGH the process group (so the line proves N ranks really took part): rank id, its
        # own wall time for the K steps, its gather time
        mine = torch.tensor([float(rank), elapsed, t_gather], dtype=torch.float64, device=dev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(x) for x in r.cpu()] for r in allr]
        tt = torch.tensor([elapsed, t_ar, t_codec, t_gather], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, t_ar, t_codec, t_gather = float(tt[0]), float(tt[1]), float(tt[2]), float(tt[3])
        if rank == 0:                         # the gather really delivered every rank's batch
            gw, gl = gathered
            assert tuple(gw.shape) == (world, B, wav.shape[1]) and bool((gl == F * ccfg.total_upsample).all())
            assert bool(torch.isfinite(gw).all())
    assert all(int(x) == F * ccfg.total_upsample for x in wl)
    assert bool(torch.isfinite(wav).all())

    G = tcfg.num_code_groups
    tokens_total = world * B * F * G * args.steps
    audio_s_total = world * B * F * (ccfg.total_upsample / 24000.0) * args.steps
    value = tokens_total / elapsed
    res = {
        "metric": "speech-tokens/sec (+ audio RTF), Qwen3-TTS-12Hz-1.7B batch=8",
        "value": round(value, 1), "unit": "speech-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.talker_dtype == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": f"Qwen3-TTS-12Hz-{args.model} dims, seeded random weights, batch={B} per GPU, ragged prompts "
                               f"{min(lens)}..{max(lens)} rows, {F} frames ({F * 0.08:.1f} s) per utterance, "
                               f"{'greedy' if args.greedy else 'sampling T=0.9 top-k=50 rep=1.05'}; prefill + AR decode "
                               f"(hipGraph={'off' if args.no_graph else 'on'}) + codec decode to 24 kHz ({args.codec_dtype})",
                   "global_batch": world * B, "frames_per_utterance": F, "parallelism": f"request-shard x{world}"},
        "rtf_x": round(audio_s_total / elapsed, 2),
        "rtf_x_per_stream": round(audio_s_total / elapsed / (world * B), 2),
        "frames_per_s": round(world * B * F * args.steps / elapsed, 1),
        "ar_ms_per_frame": round(1000 * t_ar / (args.steps * F), 4),     # prefill + first token amortised in
        "codec_ms_per_step": round(1000 * t_codec / args.steps, 2),
        "build_seconds": round(build_s, 1),
        "timed_region": "seam S2 -> device waveform: talker prefill + AR decode + codec decode"
                        + (" + gather of all ranks' waveforms on rank 0 (RCCL)" if world > 1 else "")
                        + "; prompt embeddings already in HBM (tokenisation, prompt assembly a1/f1 and D2H are outside)",
    }
    if world > 1:
        res["gather_ms_per_step"] = round(1000 * t_gather / args.steps, 3)
        res["backend"] = args.backend
        res["ranks_seen"] = sorted(int(r[0]) for r in per_rank)
        res["ms_per_step_by_rank"] = [round(1000 * r[1] / args.steps, 2) for r in sorted(per_rank)]
        res["gather_ms_per_step_by_rank"] = [round(1000 * r[2] / args.steps, 3) for r in sorted(per_rank)]
    if args.overlap_codec > 0:
        res["experiment"] = (f"--overlap-codec {args.overlap_codec}: codec packets of {args.overlap_codec} frames decoded by the state-carrying "
                             "stream decoder while the talker generates the next frames; NOT the reported configuration")
    if hostemu:
        res["INVALID"] = f"run by a test wrapper on device {device!r}: launcher test only, not a measurement"

    if rank == 0:                      # extra legs, outside the timed region: roofline at any N, cpu_baseline at N = 1
        st = talker.stats()
        wbytes = st["weight_bytes_per_frame"]
        kvb = 2.0 * B * tcfg.num_hidden_layers * 2 * tcfg.num_key_value_heads * tcfg.head_dim * (np.mean(lens) + F / 2)
        res["frame_bytes_model"] = {"weights": wbytes, "kv_avg": kvb}
        res["frame_hbm_frac_of_8TBs"] = round((wbytes + kvb) / (res["ar_ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if not args.no_roofline:
            res["roofline"] = roofline_leg(talker, emb, mask, trailing, pad, gen_kw, wbytes, args.model)
        log("roofline leg done")
        if not args.no_parity_mode and world == 1 and not hostemu:
            res["parity_mode"] = parity_mode_leg(args, tcfg, ccfg, tw_np, cw_np, lens, dev, emb, mask, trailing, pad, gen_kw, None)
            log("parity-mode leg done")
        if (not args.no_api_e2e or not args.no_configs) and world == 1:
            del talker, codec                 # (the legs below build their own engines; free the bench's first)
            torch.cuda.empty_cache()
        if not args.no_configs and world == 1 and args.model in ("1.7b", "tiny"):
            try:                              # (an extra leg must never cost the run its metric line)
                res["configs"] = configs_leg(args, tcfg, ccfg, tw_np, cw_np, dev)
            except Exception as e:            # noqa: BLE001
                res["configs"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
            log("configs leg done")
        if not args.no_configs and world == 1 and args.model in ("1.7b", "tiny"):
            try:
                res["voice_clone_prompt"] = voice_clone_prompt_leg(args, tcfg, dev)
            except Exception as e:            # noqa: BLE001
                res["voice_clone_prompt"] = {"error": f"{type(e).__name__}: {e}"}
            log("voice_clone_prompt leg done")
        if not args.no_api_e2e and world == 1:
            res["api_e2e"] = api_e2e_leg(args, tcfg, ccfg, cw_np, dev, res["ms_per_step"])
            log("api_e2e leg done")
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(tcfg, ccfg, tw_np, cw_np, lens, args.cpu_frames, args.cpu_budget_s)
            log("cpu baseline done")
            try:     # the reference's OWN modules (via oracle/ref_shims.py) cannot run on the GPU box (no /root/reference there):
                     # their timing on the build container is carried as a second, clearly labelled number
                with open(os.path.join(ROOT, "profiles", "reference_cpu_timing.json")) as f:
                    res["cpu_baseline_reference"] = json.load(f).get(args.model)
            except Exception:
                res["cpu_baseline_reference"] = None
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def clone_shard_job(args, tcfg, ccfg, dev, rank, world, dist, weights=None, keep=None):
    """BASELINE config 5 under the bench launcher, STRONG scaling: a fixed job of `--requests` voice-clone (ICL-shaped: 38
    reference frames + 16 ref-text rows in the prompt) requests of different lengths, dealt longest-first to the (rank, engine)
    bins (`sharding.engine_partition`, every rank computes the same partition), each engine running waves of `--batch` requests of
    similar length, every rank's variable-length waveforms gathered on rank 0 (`sharding.gather_waveforms_device`: the shard's one
    exchange, device to device).  A step = the whole job; the timed region is bracketed by barriers and the maximum over ranks is
    reported, as for the metric workload.

    Round 4 defaults (VERDICT r3 item 4 -- the configuration a scaling run should measure): waves of 32 (a batch-32 frame step is
    4.4 ms against 2.7 ms at batch 8: 2.4x the tokens per second) and TWO engines per GPU, each with its own weights, HIP stream
    and host thread (a frame step leaves most CUs idle: +47 % measured in round 3)."""
    import threading
    import numpy as np
    import torch
    import synth
    from qwen3_tts_amd import sharding
    from qwen3_tts_amd.codec import CodecDecoderEngine
    from qwen3_tts_amd.talker import TalkerEngine
    hostemu = not str(dev).startswith("cuda")
    NREQ, B, REF, E = args.requests, args.batch, 38, max(1, args.engines)
    rq = np.random.default_rng(5)
    text = rq.integers(16, 72, NREQ).tolist()                                   # text tokens per request
    frames = [max(2, int(round(t * args.frames / 57.0))) for t in text]         # synthetic length model: ~2.2 frames per text token at --frames 125
    parts = sharding.engine_partition(text, world, E)                           # parts[rank][engine] -> request indices
    my_waves = [sharding.waves(sorted(p, key=lambda i: -text[i]), B) for p in parts[rank]]   # similar lengths share a wave
    Fmax = max(frames)
    td = lambda w: {k: torch.from_numpy(v) for k, v in w.items()}
    tdt = torch.bfloat16 if args.talker_dtype == "bf16" else torch.float32
    cdt = torch.bfloat16 if args.codec_dtype == "bf16" else torch.float32
    # (`weights` = (talker, codec) numpy dicts a caller already holds -- the `configs` leg of the metric run; `keep`: a dict that receives the engines)
    tw, cw = (td(weights[0]), td(weights[1])) if weights is not None else (td(synth.talker_weights(tcfg, with_text=False)), td(synth.codec_weights(ccfg)))
    talkers = [TalkerEngine(tcfg, tw, weight_dtype=tdt, device=dev, max_batch=B, max_seq=12 + REF + 16 + 8 + Fmax + 8,
                            use_graph=not args.no_graph, shared_device=E > 1) for _ in range(E)]       # (engines that share a device keep the decode GEMMs: talker.py)
    codecs = [CodecDecoderEngine(ccfg, cw, compute_dtype=cdt, device=dev, max_batch=B, max_frames=min(Fmax, 300) + 25) for _ in range(E)]
    del tw, cw
    if keep is not None:
        keep.update(talkers=talkers, codecs=codecs, frames=frames)
    sup = [i for i in range(tcfg.vocab_size - 1024, tcfg.vocab_size) if i != tcfg.codec_eos_token_id]
    base = dict(suppress_tokens=sup, repetition_penalty=1.05, output_hidden_states=False, do_sample=True, top_k=50, top_p=1.0,
                temperature=0.9, subtalker_dosample=True, subtalker_top_k=50, subtalker_top_p=1.0, subtalker_temperature=0.9)

    def run_wave(w, seed, e):
        lens = [12 + REF + 16 + (text[i] % 7) for i in w]        # role + codec prefix + [ref text + text] over [ref codes]
        g = np.random.default_rng(1000 + w[0])
        emb, mask, trailing, pad = [x.to(dev) for x in synth.rand_prompt(g, tcfg, lens, max(text[i] for i in w), 0.05)]
        F = max(frames[i] for i in w)
        out = talkers[e].generate(emb, mask, trailing, pad, seed=seed, max_new_tokens=F + 1, min_new_tokens=F + 1, **base)
        codes = out.codes.clone()
        for j, i in enumerate(w):                                # each request keeps its own length (rest = -1 padding)
            codes[j, frames[i]:] = -1
        wav, wl = codecs[e].decode_padded(codes)
        return [wav[j, :int(wl[j])] for j in range(len(w))]

    def engine_loop(e, seed0, results, errors):
        try:
            torch.cuda.set_device(torch.device(dev))
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):       # per-thread current stream: copies and the codec stay off the
                for k, w in enumerate(my_waves[e]):                      # shared default stream
                    results[e].append(run_wave(w, seed0 + 64 * e + k, e))
                torch.cuda.current_stream().synchronize()
        except Exception as ex:                                          # surface worker failures on the main thread
            errors.append(ex)

    def job(seed0):
        results, errors = [[] for _ in range(E)], []
        if E == 1 or hostemu:                                            # (the emulator's SIMT state is process-global: its engines take turns)
            for e in range(E):
                engine_loop(e, seed0, results, errors)
        else:                                                            # ctypes releases the GIL inside the C calls
            th = [threading.Thread(target=engine_loop, args=(e, seed0, results, errors)) for e in range(E)]
            [t.start() for t in th]
            [t.join() for t in th]
        if errors:
            raise errors[0]
        wavs, idx = [], []
        for e in range(E):
            for w, r in zip(my_waves[e], results[e]):
                idx += list(w)
                wavs += r
        torch.cuda.synchronize()
        tg = time.perf_counter()
        if dist is not None:
            allw = sharding.gather_waveforms_device(wavs, idx, NREQ)     # device -> device (nccl = RCCL); no host round trip
        else:
            allw = [None] * NREQ
            for i, x in zip(idx, wavs):
                allw[i] = x
        torch.cuda.synchronize()
        return allw, time.perf_counter() - tg

    for i in range(max(1, args.warmup)):                         # (at least one warm-up: graph capture per wave shape, on every engine)
        for e in range(E):
            if my_waves[e]:
                run_wave(my_waves[e][0], 1 + i, e)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_gather = 0.0
    for s in range(args.steps):
        allw, tg = job(100 + 1000 * s)
        t_gather += tg
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = None
    n_mine = sum(len(p) for p in parts[rank])
    if dist is not None:
        gdev = dev if args.backend == "nccl" else "cpu"
        mine = torch.tensor([float(rank), elapsed, t_gather, float(n_mine)], dtype=torch.float64, device=gdev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = sorted([float(x) for x in r.cpu()] for r in allr)
        elapsed = max(r[1] for r in per_rank)
    if rank != 0:
        return None
    assert all(allw[i] is not None and allw[i].shape[0] == frames[i] * ccfg.total_upsample for i in range(NREQ))
    assert all(bool(torch.isfinite(allw[i]).all()) for i in range(0, NREQ, max(1, NREQ // 16)))
    G = tcfg.num_code_groups
    tok = sum(frames) * G * args.steps
    audio_s = sum(frames) * ccfg.total_upsample / 24000.0 * args.steps
    all_waves = [w for pr in parts for pe in pr for w in sharding.waves(sorted(pe, key=lambda i: -text[i]), B)]
    load = [sum(text[i] for i in pe) for pr in parts for pe in pr]
    res = {"metric": "speech-tokens/sec (+ audio RTF), Qwen3-TTS-12Hz-1.7B Base voice-clone job sharded over the GPUs (BASELINE config 5)",
           "value": round(tok / elapsed, 1), "unit": "speech-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": args.talker_dtype, "data": "synthetic",
           "config": {"workload": f"clone-shard: {NREQ} voice-clone requests (ICL-shaped prompts: 38 reference frames + 16 ref-text rows, "
                                  f"16..71 text tokens, {min(frames)}..{max(frames)} frames each), Qwen3-TTS-12Hz-{args.model} dims, seeded random "
                                  f"weights, {E} engine(s) per GPU, waves of {B}, LPT partition over (rank, engine), waveforms gathered on rank 0 (device)",
                      "requests": NREQ, "wave_batch": B, "engines_per_gpu": E, "parallelism": f"request-shard x{world} x {E} engines"},
           "wave_batch": B, "engines_per_gpu": E,
           "rtf_x": round(audio_s / elapsed, 2), "gather_ms_per_step": round(1e3 * t_gather / args.steps, 3),
           "requests_by_rank": [sum(len(pe) for pe in pr) for pr in parts],
           "rank_load_imbalance": round(max(sum(text[i] for pe in pr for i in pe) for pr in parts) / (sum(text) / world), 3),
           "engine_load_imbalance": round(max(load) / (sum(text) / (world * E)), 3),
           "padding_waste": round(1.0 - sum(frames) / sum(max(frames[i] for i in w) * len(w) for w in all_waves), 3),
           "_wave_frames": [max(frames[i] for i in w) * args.steps for w in all_waves]}
    if per_rank is not None:
        res["backend"] = args.backend
        res["ranks_seen"] = [int(r[0]) for r in per_rank]
        res["ms_per_step_by_rank"] = [round(1e3 * r[1] / args.steps, 2) for r in per_rank]
        res["gather_ms_per_step_by_rank"] = [round(1e3 * r[2] / args.steps, 3) for r in per_rank]
    return res


def configs_leg(args, tcfg, ccfg, tw_np, cw_np, dev):
    """BASELINE.json's OTHER configurations on the driver's line, outside the timed region, N = 1 only:
      config2  Tokenizer-12Hz decode-only, 10 s of random codes -> waveform, batch 1: ms + fraction of the MFMA peak (5.12 GFLOP per
               frame), in bf16 (the serving mode) and fp32 (the mode that meets RMS <= 1e-4);
      config3  0.6B dims, batch 8, 125 frames: the metric step at the small model (prefill + AR decode + codec decode);
      config4  1.7B dims, batch 32, streaming text input: first packet = call -> 4 frames of PCM on the host, p50 / p99 over 10 trials;
      config5  the 256-request voice-clone job at N = 1 (2 engines x waves of 32; `--workload clone-shard` runs it at N > 1).
    Synthetic seeded weights and prompts as everywhere; every sub-object names its roofline denominator."""
    import argparse as _ap
    import numpy as np
    import torch
    import synth
    from qwen3_tts_amd.codec import CodecDecoderEngine
    from qwen3_tts_amd.talker import TalkerEngine
    td = lambda w: {k: torch.from_numpy(v) for k, v in w.items()}
    out = {}
    t_leg = time.perf_counter()
    small = args.model == "tiny"            # (tests/hostemu/bench_emu.py: the same leg at test dims on the emulator)
    # ---- config 2
    F = 3 if small else 125
    c2 = {"workload": "Qwen3-TTS-Tokenizer-12Hz decode-only, 125 frames (10 s) of random codes -> 24 kHz waveform, batch 1", "runs": []}
    rng = np.random.default_rng(2)
    for dt, peak, name in ((torch.bfloat16, 2500.0, "bf16"), (torch.float32, 157.3, "f32")):
        if small and name == "bf16":
            continue                         # (the emulator's bf16 matrix path is slow; the leg's plumbing does not depend on the dtype)
        eng = CodecDecoderEngine(ccfg, td(cw_np), compute_dtype=dt, device=dev, max_batch=1, max_frames=F + 25)
        codes = torch.from_numpy(rng.integers(0, ccfg.codebook_size, (1, F, ccfg.num_quantizers))).to(dev)
        for _ in range(1 if small else 3):
            wav, wl = eng.decode_padded(codes)
        torch.cuda.synchronize()
        ts = []
        for _ in range(1 if small else 10):
            t0 = time.perf_counter()
            wav, wl = eng.decode_padded(codes)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        assert bool(torch.isfinite(wav).all()) and int(wl[0]) == F * ccfg.total_upsample
        ms = 1e3 * float(np.median(ts))
        tf = 5.12 * F / ms
        c2["runs"].append({"dtype": name, "ms_p50": round(ms, 3), "ms_min": round(1e3 * min(ts), 3), "tflops": round(tf, 1),
                           "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4)},
                           "rtf_x": round(F * 0.08 / (ms * 1e-3), 1)})
        del eng
    out["config2_codec_only"] = c2
    log(f"configs: config 2 done (+{time.perf_counter() - t_leg:.1f}s)")
    # ---- config 3
    t6 = tcfg if small else synth.talker_06b()
    B = 2 if small else 8
    lens = [24 + 4 * (i % 8) + 12 for i in range(B)] if not small else [5, 7]
    sdt = torch.float32 if small else torch.bfloat16
    tk = TalkerEngine(t6, td(synth.talker_weights(t6, with_text=False)), weight_dtype=sdt, device=dev, max_batch=B, max_seq=max(lens) + F + 8,
                      use_graph=not args.no_graph)
    cd = CodecDecoderEngine(ccfg, td(cw_np), compute_dtype=sdt, device=dev, max_batch=B, max_frames=F + 25)
    emb, mask, trailing, pad = [x.to(dev) for x in synth_prompt(np.random.default_rng(100), t6, lens, 1)]
    sup = [i for i in range(t6.vocab_size - 1024, t6.vocab_size) if i != t6.codec_eos_token_id]
    kw = dict(max_new_tokens=F + 1, min_new_tokens=F + 1, suppress_tokens=sup, repetition_penalty=1.05, output_hidden_states=False, do_sample=True,
              top_k=50, top_p=1.0, temperature=0.9, subtalker_dosample=True, subtalker_top_k=50, subtalker_top_p=1.0, subtalker_temperature=0.9)

    def step6(seed):
        o = tk.generate(emb, mask, trailing, pad, seed=seed, **kw)
        assert o.n_frames == F
        return cd.decode_padded(o.codes)
    step6(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K3 = 1 if small else 5
    for i in range(K3):
        wav, wl = step6(10 + i)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K3
    t0 = time.perf_counter()
    tk.generate(emb, mask, trailing, pad, seed=3, **kw)
    torch.cuda.synchronize()
    ar = time.perf_counter() - t0
    wb = tk.stats()["weight_bytes_per_frame"]
    kvb = 2.0 * B * t6.num_hidden_layers * 2 * t6.num_key_value_heads * t6.head_dim * (np.mean(lens) + F / 2)
    out["config3_06b_b8"] = {"workload": "Qwen3-TTS-12Hz-0.6B dims, batch 8, ragged prompts 36..64 rows, 125 frames, sampling; prefill + AR decode (hipGraph) + codec decode (bf16)",
                             "value": round(B * F * t6.num_code_groups / el, 1), "unit": "speech-tokens/s", "ms_per_step": round(1e3 * el, 2), "steps": K3,
                             "ar_ms_per_frame": round(1e3 * ar / F, 4), "rtf_x": round(B * F * 0.08 / el, 1),
                             "roofline": {"bound": "hbm", "what": "whole AR frame: (packed weights + average KV) bytes / ar_ms_per_frame", "achieved": round((wb + kvb) / (ar / F) / 1e9, 1),
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round((wb + kvb) / (ar / F) / 1e9 / HBM_PEAK_GBS, 4)},
                             "cp_fused": {k: tk.stats()[k] for k in ("cp_fused_active", "cp_fused_per_step")}}
    del tk, cd
    torch.cuda.empty_cache()
    log(f"configs: config 3 done (+{time.perf_counter() - t_leg:.1f}s)")
    # ---- config 5 at N = 1 (its engines, batch 32, serve config 4 too)
    a5 = _ap.Namespace(**vars(args))
    a5.requests, a5.batch, a5.engines, a5.steps, a5.warmup, a5.workload = (6, 2, 2, 1, 1, "clone-shard") if small else (256, 32, 2, 1, 1, "clone-shard")
    keep = {}
    r5 = clone_shard_job(a5, tcfg, ccfg, dev, 0, 1, None, weights=(tw_np, cw_np), keep=keep)
    wb17 = keep["talkers"][0].stats()["weight_bytes_per_frame"]
    waves_frames = r5.pop("_wave_frames", None)
    out["config5_clone_shard_n1"] = {"workload": r5["config"]["workload"], "value": r5["value"], "unit": r5["unit"], "ms_per_job": r5["ms_per_step"],
                                     "rtf_x": r5["rtf_x"], "wave_batch": a5.batch, "engines_per_gpu": 2, "padding_waste": r5["padding_waste"],
                                     "engine_load_imbalance": r5["engine_load_imbalance"], "n_gpus": 1,
                                     "note": "the N = 1 point of the strong-scaling job; `bench.py --workload clone-shard --gpus N` runs the same job over N ranks"}
    if waves_frames:
        streamed = wb17 * sum(waves_frames)                  # every frame step of every wave streams the packed weights once
        out["config5_clone_shard_n1"]["roofline"] = {"bound": "hbm", "what": "packed-weight bytes streamed by all frame steps of the job (both engines) / job time",
                                                      "achieved": round(streamed / (r5["ms_per_step"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                      "frac": round(streamed / (r5["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    log(f"configs: config 5 done (+{time.perf_counter() - t_leg:.1f}s)")
    # ---- config 4: first packet at batch 32.  ONE engine serves the 32 streams and has the device to itself (round 6: config 5's engines share
    # theirs and keep the decode GEMMs -- talker.py: shared_device; an engine alone runs the code predictor's MLP as one launch at batch 32), so
    # config 5's talkers go first (their places in the device's account with them) and a fresh engine is built; the codec engine is kept
    codec = keep["codecs"][0]
    tdt4 = keep["talkers"][0].weight_dtype
    for t_ in keep.pop("talkers"):
        t_.__del__()
    torch.cuda.empty_cache()
    talker = TalkerEngine(tcfg, td(tw_np), weight_dtype=tdt4, device=dev, max_batch=(2 if small else 32), max_seq=(64 if small else 256), use_graph=not args.no_graph)
    B, NF = (2 if small else 32), (2 if small else 4)
    text = [24 + 4 * (i % 8) for i in range(B)] if not small else [3, 4]     # text tokens, fed one per frame (streaming text input, M:2229-2232)
    lens = [32 + 12 + (i % 5) for i in range(B)] if not small else [6, 5]    # instruct (32) + role / codec prefix rows, ragged
    emb, mask, trailing, pad = [x.to(dev) for x in synth.rand_prompt(np.random.default_rng(4), tcfg, lens, max(text), 0.05)]
    sup = [i for i in range(tcfg.vocab_size - 1024, tcfg.vocab_size) if i != tcfg.codec_eos_token_id]
    kw = dict(kw, suppress_tokens=sup, max_new_tokens=NF + 1, min_new_tokens=NF + 1)
    lat, legs = [], []
    for trial in range(3 if small else 12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = talker.generate(emb, mask, trailing, pad, seed=50 + trial, **kw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        wav, wl = codec.decode_padded(o.codes[:, :NF])
        pcm = wav.cpu()                                        # the first packet is on the host here
        t2 = time.perf_counter()
        assert o.n_frames == NF and pcm.shape[1] == NF * ccfg.total_upsample
        if trial >= 2:
            lat.append(1e3 * (t2 - t0))
            legs.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    lat = np.array(lat)
    qd, kvd = tcfg.num_attention_heads * tcfg.head_dim, tcfg.num_key_value_heads * tcfg.head_dim
    talker_w = 2.0 * tcfg.num_hidden_layers * (tcfg.hidden_size * (2 * qd + 2 * kvd) + 3 * tcfg.hidden_size * tcfg.intermediate_size)   # bf16 layer weights: one pass = the prefill
    floor_b = talker_w + NF * wb17
    talker_ms = float(np.median([a for a, _ in legs]))
    out["config4_first_packet_b32"] = {"workload": "Qwen3-TTS-12Hz-1.7B dims, VoiceDesign-shaped batch of 32 (32 instruct rows + prefix), streaming text input, sampling; "
                                                   "first packet = generate() call -> 4 frames (320 ms) of PCM on the host",
                                       "trials": len(lat), "p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
                                       "min_ms": round(float(lat.min()), 3), "prefill_plus_ar_ms_p50": round(float(np.median([a for a, _ in legs])), 3),
                                       "codec_plus_d2h_ms_p50": round(float(np.median([b for _, b in legs])), 3),
                                       "cp_fused": {k: talker.stats()[k] for k in ("cp_fused_active", "cp_mlp_per_step", "ks_split_per_step")},
                                       "roofline": {"bound": "hbm", "what": "weight bytes the talker leg must stream (one pass of the talker layers for the prefill + 4 frame "
                                                                           "steps incl. the code predictor's 15 passes each) / prefill_plus_ar_ms_p50",
                                                    "achieved": round(floor_b / (talker_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                    "unit": "GB/s", "frac": round(floor_b / (talker_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    log(f"configs: config 4 done (+{time.perf_counter() - t_leg:.1f}s)")
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    return out


def voice_clone_prompt_leg(args, tcfg, dev):
    """`Qwen3TTSModel.create_voice_clone_prompt` (IM:356-458) for 8 and for 32 reference clips of 3 s: audio normalisation, the codec
    ENCODER (f3: waveform -> 16 codebooks, batched) and the SPEAKER encoder (f4: log-mel + ECAPA-TDNN; the reference loops clip by clip,
    IM:440-455 -- round 6: equal-length clips in batches of 8, `speaker_batched8_ms` beside `speaker_clip_by_clip_ms`) at the RELEASED dimensions (synth.mimi_enc_real / speaker_real; parity at these dims: tests/test_gpu_parity.py), the
    real wrapper around a model stand-in that owns the two engines (the talker is not involved in this call)."""
    import numpy as np
    import torch
    import synth
    from qwen3_tts_amd.codec import Qwen3TTSTokenizer
    from qwen3_tts_amd.model import Qwen3TTSModel
    from qwen3_tts_amd.speaker import SpeakerEncoderEngine
    small = args.model == "tiny"
    td = lambda w: {k: torch.from_numpy(v) for k, v in w.items()}
    t0 = time.time()
    enc = synth.mimi_enc_small() if small else synth.mimi_enc_real()
    spk = synth.speaker_small() if small else synth.speaker_real(tcfg.hidden_size if tcfg.hidden_size >= 1024 else 2048)
    dec = synth.codec_tiny() if small else synth.codec_real()
    n = 2048 if small else 72000
    counts = (2,) if small else (8, 32)
    tok_sd = dict(synth.codec_weights(dec))
    tok_sd.update({"encoder." + k: v for k, v in synth.mimi_enc_weights(enc).items()})
    tok_cfg = dict(synth.cfg_dict(dec), encoder_config=synth.cfg_dict(enc), encoder_valid_num_quantizers=enc.encoder_valid_num_quantizers,
                   encode_downsample_rate=enc.encode_downsample_rate, input_sample_rate=24000)
    out = {"api": "Qwen3TTSModel.create_voice_clone_prompt(ref_audio=[(waveform, 24000)] * N, ref_text=[...]) -> N VoiceClonePromptItem",
           "clip_seconds": round(n / 24000.0, 3), "runs": []}
    for dt, name in ((torch.float32, "f32"),) if small else ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        tok = Qwen3TTSTokenizer.from_state_dict(tok_cfg, td(tok_sd), device=dev, dtype=dt, max_batch=max(counts), max_frames=32)
        se = SpeakerEncoderEngine(synth.cfg_dict(spk), td(synth.speaker_weights(spk)), compute_dtype=dt, device=dev, max_batch=8, max_samples=n)

        class _BaseModel:                      # what the wrapper touches for this call (IM:356-458)
            tts_model_type = "base"
            speaker_encoder_sample_rate = 24000
            speech_tokenizer = tok
            device = tok.device

            def extract_speaker_embedding(self, audio, sr):
                return se.extract_speaker_embedding(audio, sr)

            def extract_speaker_embeddings(self, audios, sr):           # round 6: equal-length clips as batches of 8 (model.py)
                return se.embed_many(audios)
        tts = Qwen3TTSModel(_BaseModel(), _BenchProcessor(tcfg), generate_defaults={})
        for N in counts:
            clips = [(a, 24000) for a in synth.rand_audio(300 + N, N, n)]
            texts = ["reference words"] * N
            items = tts.create_voice_clone_prompt(ref_audio=clips, ref_text=texts)            # warm-up: builds the encoder engine, fills caches
            assert len(items) == N and items[0].ref_code.shape == (-(-n // enc.encode_downsample_rate), enc.encoder_valid_num_quantizers)
            assert items[0].ref_spk_embedding.shape == (spk.enc_dim,) and bool(torch.isfinite(items[0].ref_spk_embedding).all())
            torch.cuda.synchronize()
            ts = []
            for _ in range(1 if small else 5):
                ta = time.perf_counter()
                tts.create_voice_clone_prompt(ref_audio=clips, ref_text=texts)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - ta)
            wavs = torch.from_numpy(np.stack([a for a, _ in clips])).to(dev)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            tok.model._encoder.encode_padded(wavs)
            torch.cuda.synchronize()
            enc_ms = 1e3 * (time.perf_counter() - ta)
            ta = time.perf_counter()
            for i in range(N):
                se.embed(wavs[i:i + 1])
            torch.cuda.synchronize()
            spk_ms = 1e3 * (time.perf_counter() - ta)
            ta = time.perf_counter()
            for i in range(0, N, 8):
                se.embed(wavs[i:i + 8])
            torch.cuda.synchronize()
            spk_b_ms = 1e3 * (time.perf_counter() - ta)
            out["runs"].append({"dtype": name, "clips": N, "ms_per_call": round(1e3 * float(np.median(ts)), 2), "ms_min": round(1e3 * min(ts), 2),
                                "ms_per_clip": round(1e3 * float(np.median(ts)) / N, 3), "encoder_batched_ms": round(enc_ms, 2),
                                "speaker_clip_by_clip_ms": round(spk_ms, 2), "speaker_batched8_ms": round(spk_b_ms, 2),
                                "audio_seconds_per_wall_second": round(N * n / 24000.0 / float(np.median(ts)), 1)})
        del tts, tok, se
        torch.cuda.empty_cache()
    out["build_seconds"] = round(time.time() - t0, 1)
    return out


def fused_cp_report(c, fused_rec, elem_bytes=2):
    """`roofline.fused_cp_launch`.  Round 4: in passes >= 1 of the code predictor the q|k|v GEMM, the attention and the o-projection run as ONE
    launch (attention.hip: cp_attn_o_kernel) -- weight bytes that left the decode GEMM's launches and are reported beside them.  Not timed by
    the per-launch events of `roofline_leg`: durations and FETCH_SIZE bytes come from the stamped rocprofv3 passes (`fused_rec` = the "fused"
    section of profiles/pmc_traffic.json), algorithmic bytes = the two operators of a layer (`front`) or the o-projection alone (`attn_o`:
    layer 0, whose q|k|v row comes from the table)."""
    qd, kvd, H = c.cp_num_attention_heads * c.cp_head_dim, c.cp_num_key_value_heads * c.cp_head_dim, c.cp_hidden_size
    G, L = c.num_code_groups, c.cp_num_hidden_layers
    alg = {"front": ((qd + 2 * kvd) * H + H * qd) * elem_bytes, "attn_o": H * qd * elem_bytes, "mlp": 3 * c.cp_intermediate_size * H * elem_bytes}
    alg["layer_front"] = alg["front"] + alg["mlp"]; alg["layer"] = alg["attn_o"] + alg["mlp"]      # (round 6: cp_layer_kernel, both stages in one launch)
    per_frame = {"front": (G - 2) * (L - 1), "attn_o": G - 2, "mlp": (G - 2) * L, "layer_front": (G - 2) * (L - 1), "layer": G - 2}
    fl = {}
    for k, f in fused_rec.items():
        if k in alg and isinstance(f, dict) and f.get("rocprof_avg_launch_us"):
            fl[k] = {"launches_per_frame": per_frame[k], "algorithmic_bytes_per_launch": alg[k], "rocprof_avg_launch_us": f["rocprof_avg_launch_us"],
                     "frac_rocprof": round(alg[k] / (f["rocprof_avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": f.get("fetch_bytes_per_launch")}
    if fl:
        fl["weight_bytes_per_frame"] = sum(v["launches_per_frame"] * v["algorithmic_bytes_per_launch"] for v in fl.values())
        fl["kernel"] = ("cp_attn_o_kernel (front: q|k|v GEMM + attention + o-projection of a code-predictor layer in one launch; attn_o: layer 0, "
                        "whose q|k|v row comes from the table)")
    return fl


def roofline_leg(talker, emb, mask, trailing, pad, gen_kw, wbytes, model, elem_bytes=2,
                 kernel="skinny8_kernel (weight-streaming decode GEMM, batch <= 8 instantiations)"):
    """Live measurement of the dominant kernel IN THE REAL FRAME STEP (round 3): the engine runs 8 real frames eagerly and times
    every launch of the skinny weight-streaming decode GEMM of frames 1..6 on its own -- the kernel's own begin / end timestamps
    (hipExtLaunchKernelGGL events on the engine's stream: the numbers rocprofv3's kernel trace reports for the same launches).
    `achieved` = algorithmic bytes per launch (the packed weight matrix, read once) / that average duration, launch-weighted over
    all classes; `classes` gives the same per GEMM shape, `by_stack` per part of the frame (code predictor / talker layers / head).
    Round 2's number (the GEMM launches of a frame replayed in isolation as their own hipGraph) rides along as `isolated_*`."""
    names = {0: "talker_layers", 1: "code_predictor", 2: "talker_head"}
    talker.set_profile(1)
    talker.generate(emb, mask, trailing, pad, seed=7, **dict(gen_kw, max_new_tokens=9, min_new_tokens=9))
    talker.set_profile(0)
    cls_all = talker.gemm_profile()
    # stack 3 = the code predictor's fused launch (cp_attn_o_kernel: q|k|v GEMM + attention + o-projection of a layer), timed by the
    # same per-launch events in the same run (round 5); the dominant-kernel figure stays the decode GEMM's own launches
    cls = [c for c in cls_all if c["stack"] in (0, 1, 2)]
    cls_fused = [c for c in cls_all if c["stack"] in (3, 4, 5)]       # 3: cp_attn_o_kernel; 4: cp_mlp_kernel (gate|up + SwiGLU + down of a code-predictor layer); 5: cp_layer_kernel (round 6: both stages in one launch)
    frames = 6
    out = {"bound": "hbm", "kernel": kernel, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "method": "per-launch kernel begin/end timestamps (hipExtLaunchKernelGGL events) over every decode-GEMM "
                                      f"launch of {frames} real frame steps, eager launches; frac per class = N*K*{elem_bytes} B / avg duration / 8 TB/s"}
    tot_ms = sum(c["total_ms"] for c in cls)
    tot_n = sum(c["launches"] for c in cls)
    tot_b = sum(c["launches"] * c["bytes_per_launch"] for c in cls)
    if tot_n == 0 or tot_ms <= 0:
        out.update(achieved=None, frac=None, traffic=None, error="profile mode returned no launches")
        return out
    ach = tot_b / (tot_ms * 1e-3) / 1e9
    out.update(achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4), launches_per_frame=tot_n // frames,
               avg_launch_us=round(1e3 * tot_ms / tot_n, 3), algorithmic_bytes_per_launch=round(tot_b / tot_n),
               weight_bytes_per_frame_model=wbytes, weight_bytes_per_frame_timed=round(tot_b / frames))
    rows = []
    for c in sorted(cls, key=lambda c: (c["stack"], -c["bytes_per_launch"])):
        us = 1e3 * c["total_ms"] / c["launches"]
        gbs = c["bytes_per_launch"] / (us * 1e-6) / 1e9
        rows.append({"stack": names.get(c["stack"], str(c["stack"])), "N": c["N"], "K": c["K"], "launches_per_frame": c["launches"] // frames,
                     "MB": round(c["bytes_per_launch"] / 1e6, 2), "avg_us": round(us, 3), "min_us": round(c["min_us"], 3),
                     "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)})
    out["classes"] = rows
    by = {}
    for c in cls:
        a = by.setdefault(names.get(c["stack"], str(c["stack"])), [0.0, 0, 0.0])
        a[0] += c["total_ms"]; a[1] += c["launches"]; a[2] += c["launches"] * c["bytes_per_launch"]
    out["by_stack"] = {k: {"launches_per_frame": v[1] // frames, "avg_us": round(1e3 * v[0] / v[1], 3),
                           "frac": round(v[2] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for k, v in by.items()}
    if cls_fused:
        # measured IN THIS RUN: the fused launches' own begin / end timestamps, and the like-for-like figure over
        # EVERY weight-streaming launch of the frame (decode GEMMs + fused launches)
        fl = {}
        for c in cls_fused:
            us = 1e3 * c["total_ms"] / c["launches"]
            cpc = talker.config
            key = ("mlp" if c["stack"] == 4 else ("front" if c["N"] > cpc.cp_hidden_size else "attn_o")) if c["stack"] != 5 else \
                  ("layer_front" if c["N"] > cpc.cp_hidden_size + 3 * cpc.cp_intermediate_size else "layer")
            fl[key] = {"launches_per_frame": c["launches"] // frames, "algorithmic_bytes_per_launch": round(c["bytes_per_launch"]),
                       "avg_us": round(us, 3), "min_us": round(c["min_us"], 3),
                       "frac": round(c["bytes_per_launch"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        fl["weight_bytes_per_frame"] = round(sum(c["launches"] * c["bytes_per_launch"] for c in cls_fused) / frames)
        fl["kernel"] = ("cp_layer_kernel (round 6; layer_front: q|k|v GEMM + attention + o-projection + RMSNorm + gate|up + SwiGLU + down + residuals of a code-predictor "
                        "layer in ONE launch; layer: layer 0, whose q|k|v row comes from the table) where the engine holds the device's layer place; otherwise "
                        "cp_attn_o_kernel (front: q|k|v GEMM + attention + o-projection of a code-predictor layer in one launch; attn_o: layer 0, "
                        "whose q|k|v row comes from the table) and cp_mlp_kernel (mlp: RMSNorm + gate|up GEMM + SwiGLU + down GEMM + residual of a layer in "
                        "one launch); per-launch events of this run")
        out["fused_cp_launch"] = fl
        ms_all = tot_ms + sum(c["total_ms"] for c in cls_fused)
        b_all = tot_b + sum(c["launches"] * c["bytes_per_launch"] for c in cls_fused)
        out["frac_all_weight_launches"] = round(b_all / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["weight_launches_per_frame"] = (tot_n + sum(c["launches"] for c in cls_fused)) // frames
        out["weight_bytes_per_frame_timed_all"] = round(b_all / frames)
    st_f = talker.stats()
    out["cp_fused"] = {k: st_f[k] for k in ("cp_fused_active", "cp_fused_capacity", "cp_fused_per_step", "cp_mlp_per_step", "cp_layer_per_step", "ks_split_per_step", "cp_fused_giveups") if k in st_f}
    if elem_bytes != 2:            # (the parity-mode leg: no PMC pass and no isolated replay for the fp32 engine)
        out["traffic"] = None
        return out
    # HBM bytes per launch: the round's own rocprofv3 --pmc FETCH_SIZE pass (separate pass, x1024 x2: MI355X_MICROARCH.md), reduced by
    # tools/pmc_traffic.py to a traffic / algorithmic ratio over the skinny8_kernel dispatches and STAMPED with a digest of the kernel
    # and engine sources: a tree whose digest differs reports `traffic: null` instead of a stale number.  The same
    # file carries `frac_rocprof`, the fraction rocprofv3's kernel trace of the bench command gives for the same launches.
    traffic, traffic_src = None, None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_traffic
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tj = json.load(f)
        rec = tj.get(model, {})
        if tj.get("kernel_digest") != pmc_traffic.kernel_digest():
            traffic_src = (f"profiles/pmc_traffic.json is STALE (taken on kernel digest {tj.get('kernel_digest')}, this tree is "
                           f"{pmc_traffic.kernel_digest()}): re-run the PMC pass (tools/pmc_traffic.py)")
        elif "ratio_traffic_over_algorithmic" in rec:
            traffic = round(out["algorithmic_bytes_per_launch"] * rec["ratio_traffic_over_algorithmic"])
            traffic_src = (f"builder-run rocprofv3 --pmc FETCH_SIZE pass on this tree's kernels ({rec.get('source')}; digest "
                           f"{tj['kernel_digest']}): traffic / algorithmic = {rec['ratio_traffic_over_algorithmic']} over {rec['dispatches']} "
                           "skinny8_kernel dispatches; not measured by this run")
            if "frac_rocprof" in rec:
                out["frac_rocprof"] = rec["frac_rocprof"]
                out["rocprof_avg_launch_us"] = rec["rocprof_avg_launch_us"]
                out["frac_live_over_rocprof"] = round(out["frac"] / rec["frac_rocprof"], 4)
            if "fused" in rec and out["weight_bytes_per_frame_timed"] < wbytes:
                fl = fused_cp_report(talker.config, rec["fused"], elem_bytes)
                if fl:                         # the stamped rocprofv3 / FETCH_SIZE pass rides beside the live numbers
                    for k in ("front", "attn_o", "mlp"):
                        if k in fl and k in out.get("fused_cp_launch", {}):
                            out["fused_cp_launch"][k].update(rocprof_avg_launch_us=fl[k]["rocprof_avg_launch_us"], frac_rocprof=fl[k]["frac_rocprof"],
                                                             traffic=fl[k]["traffic"])
                    if "fused_cp_launch" not in out:
                        out["fused_cp_launch"] = fl
    except Exception as e:
        traffic_src = f"unavailable ({type(e).__name__}: {e})"
    out["traffic"], out["traffic_source"] = traffic, traffic_src
    out["frac_live"] = out["frac"]
    try:          # continuity with rounds 1-2: the same launches replayed in isolation (no attention / sampler / glue nodes between them)
        talker.set_profile(2)
        talker.generate(emb, mask, trailing, pad, seed=7, **dict(gen_kw, max_new_tokens=9, min_new_tokens=9))
        st = talker.stats()
        if st["gemm_launches_last"] > 0 and st["gemm_ms_last"] > 0:
            us = 1e3 * st["gemm_ms_last"] / st["gemm_launches_last"]
            out["isolated_avg_launch_us"] = round(us, 3)
            out["isolated_frac"] = round(wbytes / st["graph_nodes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    finally:
        talker.set_profile(0)
    return out


def parity_mode_leg(args, tcfg, ccfg, tw_np, cw_np, lens, dev, emb, mask, trailing, pad, gen_kw, prefill_ms_bf16):
    """The configuration that MEETS north_star's parity bar (bit-exact codes under greedy decode, waveform RMS <= 1e-4) is the
    fp32 mode -- exact-fp32 MFMA in the talker, fp32 codec.  This leg gives it a driver-visible number: the same step with both
    engines in fp32 (same prompts, same sampling), timed outside the reported region."""
    import torch
    from qwen3_tts_amd.codec import CodecDecoderEngine
    from qwen3_tts_amd.talker import TalkerEngine
    B, F = args.batch, args.frames
    td = lambda w: {k: torch.from_numpy(v) for k, v in w.items()}
    t32 = TalkerEngine(tcfg, td(tw_np), weight_dtype=torch.float32, device=dev, max_batch=B, max_seq=max(lens) + F + 8, use_graph=not args.no_graph)
    c32 = CodecDecoderEngine(ccfg, td(cw_np), compute_dtype=torch.float32, device=dev, max_batch=B, max_frames=min(F, 300) + 25)
    out = t32.generate(emb, mask, trailing, pad, seed=1, **gen_kw)            # warm-up: graph capture, allocator
    c32.decode_padded(out.codes)
    torch.cuda.synchronize()
    ta = time.perf_counter()
    out = t32.generate(emb, mask, trailing, pad, seed=2, **gen_kw)
    torch.cuda.synchronize()
    tb = time.perf_counter()
    wav, wl = c32.decode_padded(out.codes)
    torch.cuda.synchronize()
    tc = time.perf_counter()
    assert out.n_frames == F and bool(torch.isfinite(wav).all())
    roof = None
    if not args.no_roofline:      # the parity mode's own roofline: every fp32 decode-GEMM launch of 6 real frame steps, per class
        roof = roofline_leg(t32, emb, mask, trailing, pad, gen_kw, t32.stats()["weight_bytes_per_frame"], args.model, elem_bytes=4,
                            kernel="skinny8_f32_kernel (exact-fp32 weight-streaming decode GEMM, v_mfma_f32_16x16x4_f32, batch <= 8)")
    del t32, c32
    torch.cuda.empty_cache()
    return {"roofline": roof,
            "talker_f32_ms_per_frame": round(1e3 * (tb - ta) / F, 4), "codec_f32_ms_per_step": round(1e3 * (tc - tb), 2),
            "step_ms": round(1e3 * (tc - ta), 2), "speech_tokens_per_s": round(B * F * tcfg.num_code_groups / (tc - ta), 1),
            "what": "exact-fp32 talker (v_mfma_f32_16x16x4_f32 chain, bit-exact greedy codes vs the reference goldens) + fp32 codec "
                    "(waveform RMS 7.4e-6 vs the reference at real dims): the mode the parity bar is proven in, same workload, 1 step"}


class _BenchProcessor:
    """Deterministic stand-in for the HF text tokenizer (no Qwen vocabulary offline): one id per 4 characters of the body, the chat
    template's special strings mapped to the ids the model config names (as the reference's processor does, IM:263-285)."""

    def __init__(self, t):
        self.t = t

    def __call__(self, text=None, return_tensors="pt", padding=True, **kw):
        import torch
        t = self.t
        a, u, n = 77, 78, 198
        body = text
        for tag in ("<|im_start|>assistant\n", "<|im_start|>user\n", "<|im_end|>\n", "<|im_start|>", "<|im_end|>"):
            body = body.replace(tag, "")
        V = int(t.text_vocab_size)
        lo = min(1000, V // 2)                         # (plain text ids; the special ids sit outside this range at real dims)
        toks = [lo + (sum(ord(ch) * (i + 1) for i, ch in enumerate(body[k:k + 4])) % min(40000, V - lo)) for k in range(0, len(body), 4)]
        if text.startswith("<|im_start|>user"):
            ids = [t.im_start_token_id, u, n] + toks + [t.im_end_token_id, n]
        else:
            ids = [t.im_start_token_id, a, n] + toks + [t.im_end_token_id, n, t.im_start_token_id, a, n]
        return {"input_ids": torch.tensor([ids])}


def api_e2e_leg(args, tcfg, ccfg, cw_np, dev, s2_ms_per_step):
    """The step at the API the north_star says to keep (IM:732-840): `Qwen3TTSModel.generate_custom_voice(texts, speakers,
    language=...)` from Python strings to host numpy waveforms -- tokenise, prompt assembly on the device (a1 / f1), prefill + AR
    decode, codec decode, D2H and list building -- on the same model dims, batch, sampling and frame count as the S2 step above,
    timed end to end on the host clock.  The parts that are NOT in the S2 number are timed on their own so that the gap is itemised."""
    import numpy as np
    import torch
    import synth
    from qwen3_tts_amd.codec import Qwen3TTSTokenizer
    from qwen3_tts_amd.model import Qwen3TTSForConditionalGeneration, Qwen3TTSModel
    B, F = args.batch, args.frames
    td = lambda w: {k: torch.from_numpy(v) for k, v in w.items()}
    t0 = time.time()
    tw = synth.talker_weights(tcfg, with_text=True)                 # + text embedding (151 936 rows at 1.7B dims) and projection
    cfgd = dict(synth.cfg_dict(tcfg), tts_model_type="custom_voice", tts_model_size="1b7", tokenizer_type="12hz")
    ntext = [24 + 4 * (i % 8) for i in range(B)]                     # text tokens per request, as the S2 step's prompt rows
    model = Qwen3TTSForConditionalGeneration(cfgd, td(tw), device=dev, dtype=torch.bfloat16 if args.talker_dtype == "bf16" else torch.float32, max_batch=B,
                                             max_seq=max(ntext) + 32 + F + 8, use_graph=not args.no_graph)
    del tw
    cdt = torch.bfloat16 if args.codec_dtype == "bf16" else torch.float32
    model.load_speech_tokenizer(Qwen3TTSTokenizer.from_state_dict(synth.cfg_dict(ccfg), td(cw_np), device=dev, dtype=cdt, max_batch=B,
                                                                  max_frames=min(F, 300) + 25))
    tts = Qwen3TTSModel(model, _BenchProcessor(tcfg), generate_defaults={})
    build_s = time.time() - t0
    rng = np.random.default_rng(7)
    alphabet = np.array(list("abcdefghijklmnopqrstuvwxyz    ,."))
    texts = ["".join(rng.choice(alphabet, 4 * n)) for n in ntext]   # 96..208 characters ("128-char synthetic prompts", BASELINE config 3)
    speakers = [sorted(tcfg.spk_id)[i % len(tcfg.spk_id)] for i in range(B)]
    langs = [sorted(tcfg.codec_language_id)[i % len(tcfg.codec_language_id)] for i in range(B)]
    # a fixed number of frames, as in the S2 step: the stop id handed to generate() is one the suppress list never lets through
    # (M:2059-2063 suppresses the top 1024 ids except the real EOS); a row that samples the real EOS id is still cut there (M:2283-2289)
    never = tcfg.vocab_size - 1
    assert never != tcfg.codec_eos_token_id
    kw = dict(language=langs, max_new_tokens=F + 1, eos_token_id=never) if not args.greedy else dict(
        language=langs, max_new_tokens=F + 1, eos_token_id=never, do_sample=False, subtalker_dosample=False)
    for i in range(max(1, args.warmup)):
        tts.generate_custom_voice(texts, speakers, seed=500 + i, **kw)
    torch.cuda.synchronize()
    K = max(3, min(args.steps, 10))
    ts, frames_out = [], []
    for i in range(K):
        ta = time.perf_counter()
        wavs, sr = tts.generate_custom_voice(texts, speakers, seed=600 + i, **kw)
        ts.append(time.perf_counter() - ta)            # the call returns host numpy arrays: nothing is left in flight
        frames_out.append(sum(w.shape[0] for w in wavs) // ccfg.total_upsample)
        assert sr == 24000 and len(wavs) == B and all(isinstance(w, np.ndarray) and w.dtype == np.float32 for w in wavs)
    ms = 1e3 * float(np.mean(ts))
    # the parts the S2 step does not contain, each on its own
    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t) / n, r
    tok_ms, ids = timed(lambda: tts._tokenize_texts([tts._build_assistant_text(x) for x in texts]))
    asm_ms, _ = timed(lambda: model.assemble_prompts(ids, langs, speakers, [None] * B, True))
    codes = torch.randint(0, ccfg.codebook_size, (B, F, ccfg.num_quantizers), device=dev)
    dec_ms, _ = timed(lambda: model.speech_tokenizer.decode([{"audio_codes": c} for c in codes]))
    pad_ms, _ = timed(lambda: model.speech_tokenizer.model.decoder.decode_padded(codes))
    G = tcfg.num_code_groups
    tok_s = B * F * G / (ms * 1e-3)
    out = {"api": "Qwen3TTSModel.generate_custom_voice(texts[8], speakers[8], language=...) -> (list of host numpy waveforms, 24000)",
           "ms_per_call": round(ms, 2), "ms_min": round(1e3 * min(ts), 2), "calls": K,
           "speech_tokens_per_s": round(tok_s, 1), "rtf_x": round(B * F * 0.08 / (ms * 1e-3), 2),
           "frames_generated_per_row": F, "frames_returned_mean": round(float(np.mean(frames_out)) / B, 2),
           "s2_ms_per_step": s2_ms_per_step, "gap_ms": round(ms - s2_ms_per_step, 2), "gap_pct": round(100.0 * (ms / s2_ms_per_step - 1.0), 2),
           "not_in_s2_ms": {"tokenise (stand-in processor)": round(tok_ms, 3), "assemble_prompts (build_prompt_plan + text_embed + assemble_rows)": round(asm_ms, 3),
                            "wrapper decode: D2H + per-request numpy list, over decode_padded": round(dec_ms - pad_ms, 3)},
           "build_seconds": round(build_s, 1),
           "note": "same dims / batch / sampling / frame count as the S2 step; text 24..52 tokens per request (96..208 characters); prompt rows come from the "
                   "real chat template + codec prefix instead of synthetic embeddings; EOS trimming (M:2283-2289) applies, so a row may return fewer frames than were generated"}
    del tts, model
    torch.cuda.empty_cache()
    return out


def _host_threads():
    """Cores this process may actually run on (cgroup/affinity aware) -- os.cpu_count() can report the whole
    host and oversubscribing torch's thread pool makes the CPU leg pathologically slow."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:  # cgroup v2 cpu.max = "<quota> <period>"
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(tcfg, ccfg, tw_np, cw_np, lens, n_frames, budget_s):
    """The CPU oracle (kind "port": torch fp32 restatement of the reference, oracle/*_ref.py) on a bounded sample
    of the same workload: same batch and prompts, `n_frames` frames instead of 125, then codec decode of those
    frames.  Threads = the cores this process is allowed to use.  The AR loop is cut short if it would exceed
    the wall-clock budget (the sample actually run is reported)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))      # the oracle is reachable from this leg only
    import codec_ref
    import talker_ref
    nthreads = _host_threads()
    torch.set_num_threads(nthreads)
    tw = {k: torch.from_numpy(v) for k, v in tw_np.items()}
    cw = {k: torch.from_numpy(v) for k, v in cw_np.items()}
    rng = np.random.default_rng(100)
    emb, mask, trailing, pad = synth_prompt(rng, tcfg, lens, 1)
    sp = talker_ref.SamplingParams()
    gen = torch.Generator().manual_seed(0)
    B = len(lens)
    t0 = time.perf_counter()
    with torch.no_grad():
        # probe: prefill + 1 frame, to size the sample to the budget
        r = talker_ref.talker_generate(tw, tcfg, emb, mask, trailing, pad, max_new_tokens=2, min_new_tokens=2, sp=sp, generator=gen)
        t_probe = time.perf_counter() - t0
        log(f"cpu probe (prefill + 1 frame, {nthreads} threads): {t_probe:.1f}s")
        if t_probe * (1 + 0.5 * n_frames) > budget_s:
            n_frames = max(1, int((budget_s / t_probe - 1) / 0.5))
        t0 = time.perf_counter()
        r = talker_ref.talker_generate(tw, tcfg, emb, mask, trailing, pad, max_new_tokens=n_frames + 1,
                                       min_new_tokens=n_frames + 1, sp=sp, generator=gen)
        t_ar = time.perf_counter() - t0
        wavs = codec_ref.model_decode(cw, ccfg, r["codes"])
    dt = time.perf_counter() - t0
    toks = B * r["codes"].shape[1] * tcfg.num_code_groups
    model = "n/a"
    try:
        with open("/proc/cpuinfo") as f:
            model = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": round(toks / dt, 1), "unit": "speech-tokens/s", "cores": nthreads, "kind": "port",
            "sample": f"same batch ({B} prompts), prefill + {r['codes'].shape[1]} frames + codec decode of those frames, "
                      f"fp32 torch CPU oracle, {dt:.1f} s wall ({t_ar:.1f} s AR)",
            "rtf_x": round(B * r["codes"].shape[1] * 0.08 / dt, 3), "cpu_model": model, "os_cpu_count": os.cpu_count()}


if __name__ == "__main__":
    main()
