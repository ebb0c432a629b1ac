"""bench.py -- A3T masked-mel training-step throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]         (N=1; N>1 without a launcher: bench.py starts the N
                                                             ranks itself, one process per GPU, like the reference's
                                                             mp.spawn in espnet2/tasks/abs_task.py:1026-1045)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward + loss + full backward + (RCCL gradient all-reduce if N>1) + grad-norm/clip/Adam
of one synthetic batch that is already resident in HBM when the timed region
starts.  Workload = BASELINE.json configs[1]: 6 enc + 6 dec Conformer blocks ("12L"), d=384,
B=32/GPU, T_mel=1000, T_phn=120, bf16 MFMA compute with fp32 accumulation/master weights.
Prints ONE JSON line on rank 0.  The CPU oracle is imported only for the cpu_baseline leg.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK = {"bf16": 2500.0, "f32": 157.3}   # TFLOP/s dense, MI355X_MICROARCH.md
HBM_PEAK = 8000.0                            # GB/s
TRAFFIC_FILE = "r06_hbm_traffic_per_kernel.json"   # this round's PMC summary (tools/r06_profiles.sh)


def live_hbm_traffic(kernel_name, args, timeout=120):
    """HBM bytes per launch of `kernel_name`, measured NOW: FETCH_SIZE and WRITE_SIZE in two SEPARATE rocprofv3 --pmc passes
    (with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes; FETCH_SIZE x2 on gfx950) over one warm-up + one
    training step of this same command in a child process.  (bytes, fetch, write, note); bytes is None when rocprofv3 is not
    there, a pass fails or times out -- the caller then falls back to the round's tracked file and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, None, None, "rocprofv3 is not on PATH"
    per = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="a3t_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--compute", args.compute, "--batch", str(args.batch),
               "--tmel", str(args.tmel), "--tphn", str(args.tphn), "--blocks", str(args.blocks), "--no-cpu-baseline",
               "--no-vocoder", "--no-collate", "--no-kernel-profile", "--no-c4", "--no-live-traffic"]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            n, tot = 0, 0.0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == c and kernel_name in r.get("Kernel_Name", ""):
                        n += 1
                        tot += float(r["Counter_Value"])
            if not n:
                return None, None, None, f"{c} pass: no counter rows for {kernel_name}"
            per[c] = tot / n * 1024.0
        except (subprocess.SubprocessError, OSError, ValueError, KeyError) as e:
            return None, None, None, f"{c} pass failed: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = 2.0 * per["FETCH_SIZE"], per["WRITE_SIZE"]
    return fetch + write, fetch, write, "measured in this run"


def fwd_flops_per_step(c, B, Tm, Tp):
    """Algorithmic forward FLOPs (SURVEY §8d / BASELINE.md §4)."""
    T = Tm + Tp
    d, ff = c.adim, c.ff

    def blk(K):
        return 8 * d * ff * c.ff_kernel + 8 * d * d + 6 * T * d + 6 * d * d + 2 * d * K

    blocks = B * T * (c.enc_blocks * blk(c.enc_kernel) + c.dec_blocks * blk(c.dec_kernel))
    blocks += (c.enc_blocks + c.dec_blocks) * 2 * d * d * T
    n, ch, k = c.postnet_layers, c.postnet_chans, c.postnet_filts
    post = 2 * k * (c.odim * ch + (n - 2) * ch * ch + ch * c.odim) if n >= 2 else 0
    head = B * Tm * (2 * c.idim * d + 2 * d * c.odim + post)
    return blocks + head


def build_trainer(cfg, device, compute, world, dropout=True, comm_dtype=torch.float32):
    """A3TTrainer = the step body of Trainer.train_one_epoch (espnet2/train/trainer.py:528-703) on flat
    buffers; recipe init except BatchNorm gamma = 1 so that no GEMM runs on an all-zero operand."""
    from a3t_amd.init import xavier_init_
    from a3t_amd.params import ParamStore
    from a3t_amd.trainer import A3TTrainer
    store = ParamStore(cfg, device)
    xavier_init_(store, seed=0, bn_gamma=1.0)
    return A3TTrainer(cfg, store, compute=compute, lr=1.0, warmup_steps=4000, grad_clip=1.0, dropout=dropout,
                      comm_dtype=comm_dtype)


def c4_leg(dev, compute, steps=8, warmup=3):
    """BASELINE.json configs[3] on ONE GPU: LibriTTS multi-speaker A3T, 6+6 blocks d=512 H=4 (d_k=128) ff=2048, x-vector (512-d)
    conditioning, B=16, T_mel=1600, T_phn=200; the same step (fwd + bwd + clip + Adam, recipe dropout) as the headline."""
    from a3t_amd.collate import synthetic_batch
    from a3t_amd.config import config_c4
    cfg = config_c4()
    B, Tm, Tp = 16, 1600, 200
    tr = build_trainer(cfg, dev, compute, 1)
    batch = synthetic_batch(cfg, B, Tm, Tp, seed=4321, device=dev)
    for _ in range(warmup):
        tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    fl = 3.0 * fwd_flops_per_step(cfg, B, Tm, Tp)
    plan = tr.engine._ffn_plan(B * (Tm + Tp))
    out = dict(workload="LibriTTS multi-speaker A3T train step: 6+6 Conformer blocks d=512 H=4 ff=2048(k3) + x-vector (512) add, "
                        "B=16, T_mel=1600, T_phn=200, fwd+bwd+clip+Adam, recipe dropout",
               ms_per_step=ms, frames_per_s=B * Tm / (ms * 1e-3), steps=steps, params=tr.store.n_params,
               algorithmic_tflop_per_step=fl / 1e12, step_tflops=fl / (ms * 1e-3) / 1e12,
               frac=fl / (ms * 1e-3) / 1e12 / MFMA_PEAK["bf16"], final_loss=float(loss),
               ffn_on_8phase_gemm=dict(keep_bit_protocol=bool(plan[0]), conv1_data_gradient=bool(plan[1])),
               hbm_allocated_gb=torch.cuda.max_memory_allocated(dev) / 1e9)
    del tr, batch
    torch.cuda.empty_cache()
    return out


def vocoder_rtf(dev, B=8, Tf=1000, reps=3, cpu=False):
    """BASELINE.json configs[4]: ParallelWaveGAN v1 (30 blocks, 64/128/64 ch, hop 300 = 4*5*3*5) mel -> wav
    for B utterances of Tf frames (12.5 s each at 24 kHz); RTF = wall / audio seconds.  fp32 MFMA GEMMs."""
    import numpy as np
    from a3t_amd.vocoder import ParallelWaveGANGeneratorHIP
    rs = np.random.RandomState(0)
    shapes = {"first_conv.weight": (64, 1, 1), "first_conv.bias": (64,), "upsample_net.conv_in.weight": (80, 80, 5),
              "last_conv_layers.1.weight": (64, 64, 1), "last_conv_layers.1.bias": (64,),
              "last_conv_layers.3.weight": (1, 64, 1), "last_conv_layers.3.bias": (1,)}
    for i, sc in enumerate((4, 5, 3, 5)):
        shapes[f"upsample_net.upsample.up_layers.{2 * i + 1}.weight"] = (1, 1, 1, 2 * sc + 1)
    for l in range(30):
        p = f"conv_layers.{l}."
        shapes.update({p + "conv.weight": (128, 64, 3), p + "conv.bias": (128,), p + "conv1x1_aux.weight": (128, 80, 1),
                       p + "conv1x1_out.weight": (128, 64, 1), p + "conv1x1_out.bias": (128,)})
    state = {}
    for k, s in shapes.items():
        fan = int(np.prod(s[1:])) if len(s) > 1 else 1
        state[k] = (rs.standard_normal(s) * (1.0 / np.sqrt(fan) if len(s) > 1 else 0.05)).astype(np.float32)
        if "up_layers" in k:
            state[k] = np.abs(state[k]) / np.abs(state[k]).sum()
    voc = ParallelWaveGANGeneratorHIP(state, device=dev)
    c = torch.from_numpy((rs.standard_normal((B, Tf, 80)) * 1.5 - 4.0).astype(np.float32)).to(dev)
    z = torch.randn(B, Tf * 300, 1, device=dev)
    voc.inference(c, z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        wav = voc.inference(c, z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    audio_s = B * Tf * 300 / 24000.0
    flops = 2.60e6 * B * Tf * 300
    res = dict(metric="vocoder RTF", rtf=dt / audio_s, ms=dt * 1e3, audio_seconds=audio_s, samples_per_s=B * Tf * 300 / dt,
               tflops=flops / dt / 1e12, frac=flops / dt / 1e12 / MFMA_PEAK["f32"], peak_tflops=MFMA_PEAK["f32"], dtype="f32",
               finite=bool(torch.isfinite(wav).all()),
               workload=f"ParallelWaveGAN v1 generator, B={B} x {Tf} frames, hop 300, 24 kHz")
    if cpu:
        # CPU baseline + parity ("own CPU restatement, 1 utterance of 400 frames"): the oracle's pwg_forward on
        # the host cores with the same weights / mel / noise; the same utterance through the HIP generator for the error
        try:
            from oracle import a3t_oracle as O
            Tc = 400
            torch.set_num_threads(min(os.cpu_count() or 1, 32))
            p = {k: torch.from_numpy(v) for k, v in state.items()}
            c1, z1 = c[:1, :Tc].cpu(), z[:1, :Tc * 300].cpu()
            t0 = time.perf_counter()
            with torch.no_grad():
                ref = O.pwg_forward(p, c1[0].t()[None], z1[0].t()[None], O.PWGConfig())[0, 0]
            tc = time.perf_counter() - t0
            got = voc.inference(c1.to(dev), z1.to(dev)).reshape(-1).cpu()
            res["cpu_baseline"] = dict(kind="port", cores=torch.get_num_threads(), rtf=tc / (Tc * 300 / 24000.0),
                                       samples_per_s=Tc * 300 / tc, sample=f"oracle pwg_forward fp32, 1 utterance x {Tc} frames ({tc:.1f} s)")
            res["wav_max_abs_err_vs_oracle"] = float((got - ref).abs().max())
            res["wav_scale"] = float(ref.abs().max())
        except Exception as e:  # noqa: BLE001
            res["cpu_baseline"] = dict(kind="port", value=None, note=type(e).__name__ + ": " + str(e)[:80])
    return res


def infill_leg(dev, B=8, Tm=1000, Tp=120, span=(400, 600), reps=5, cpu=False):
    """BASELINE.json configs[4] front half: teacher-forced span infill (ESPnetMLMModel.inference, sedit_model.py:239-284)
    of B utterances with the reference-yaml model (4+4 blocks), eval-mode engine, bf16 compute; one forward per batch,
    the predicted span [s, e) replaces the masked frames.  Reported next to the vocoder so that
    pipeline RTF = (infill + vocoder) / audio seconds."""
    from a3t_amd.collate import synthetic_batch
    from a3t_amd.config import A3TConfig
    from a3t_amd.engine import MLMEngine
    from a3t_amd.init import xavier_init_
    from a3t_amd.params import ParamStore
    c = A3TConfig()                                   # reference yaml: 4 + 4 blocks, d = 384
    store = ParamStore(c, dev)
    xavier_init_(store, seed=0, bn_gamma=1.0)
    eng = MLMEngine(c, store, compute="bf16", training=False)
    batch = synthetic_batch(c, B, Tm, Tp, seed=99, device=dev)
    batch["masked_position"][:] = False
    batch["masked_position"][:, span[0]:span[1]] = True
    eng.forward(batch, need_grad=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = eng.forward(batch, need_grad=False)
        gen = [torch.cat([batch["speech"][b, :span[0]], out["after"][b, span[0]:span[1]].float(), batch["speech"][b, span[1]:]])
               for b in range(B)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res = dict(ms=dt * 1e3, utterances=B, frames=B * Tm, finite=bool(torch.isfinite(torch.stack(gen)).all()),
               workload=f"teacher-forced infill of span [{span[0]}, {span[1]}) in {B} x {Tm} frames, 4+4 blocks, eval")
    if cpu:
        # BASELINE configs[4] "mel-L1 vs reference": the oracle's teacher-forced infill (fp32, host) of ONE sampled utterance
        # with the same weights against the bf16 device result -- mean |difference| over the 200 x 80 generated mel bins
        try:
            from oracle import a3t_oracle as O
            oc = O.A3TConfig()
            p = {k: v.detach().cpu() for k, v in store.state_dict().items()}
            b1 = {k: v[2:3].cpu() for k, v in batch.items()}
            t0 = time.perf_counter()
            with torch.no_grad():
                ref = O.inference_splice(p, b1, oc, span)
            tc = time.perf_counter() - t0
            d = (gen[2].cpu() - ref)[span[0]:span[1]]
            res["mel_l1_vs_oracle"] = float(d.abs().mean())
            res["mel_max_abs_err_vs_oracle"] = float(d.abs().max())
            res["mel_scale"] = float(ref[span[0]:span[1]].abs().max())
            res["mel_rms_err_vs_oracle"] = float(d.pow(2).mean().sqrt())
            res["mel_max_rel"] = res["mel_max_abs_err_vs_oracle"] / max(1.0, res["mel_scale"])
            res["mel_rms_rel"] = res["mel_rms_err_vs_oracle"] / max(1.0, res["mel_scale"])
            try:     # the yardstick: what the REFERENCE loses on its own reference-yaml model under bf16 autocast (tracked fixture)
                import numpy as np
                r16 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "e2e_bf16ref.npz"))
                res["reference_under_bf16_autocast_refyaml_after"] = dict(max_rel=float(r16["refyaml.after.err_max"]),
                                                                          rms_rel=float(r16["refyaml.after.err_rms"]),
                                                                          source="tests/golden/e2e_bf16ref.npz (tracked fixture, not measured in this run)")
            except Exception:  # noqa: BLE001
                pass
            res["cpu_baseline"] = dict(kind="port", cores=torch.get_num_threads(), ms_per_utterance=tc * 1e3,
                                       sample="oracle inference_splice fp32, 1 utterance")
        except Exception as e:  # noqa: BLE001
            res["mel_l1_vs_oracle"] = None
            res["cpu_note"] = type(e).__name__ + ": " + str(e)[:80]
    return res


def collate_leg(dev, B=32, Tm=1000, Tp=120, reps=3):
    """the batch construction the reference runs in a DataLoader worker (MLMCollateFn: pad, STFT ->
    mel -> log10, align -> frames, span masks, segment ids) with the feature extraction on the GPU.  Input = host
    waveforms (the PCIe copy of 4 B/sample is inside the timed region); CPU figure = the oracle's collate, 1 rep."""
    import numpy as np
    from a3t_amd.collate import MLMCollateFn
    from a3t_amd.features import LogMelFbank
    rs = np.random.RandomState(5)
    hop, fs = 300, 24000
    data = []
    for i in range(B):
        n = hop * (Tm - 1)
        cuts = np.sort(rs.choice(np.arange(1, Tm - 1), Tp - 1, replace=False))
        st = np.concatenate([[0], cuts]).astype(np.float32) * hop / fs + 1e-4
        en = np.concatenate([cuts, [Tm - 1]]).astype(np.float32) * hop / fs + 1e-4
        data.append((f"u{i}", dict(speech=(0.1 * rs.standard_normal(n)).astype(np.float32),
                                   text=rs.randint(2, 70, size=Tp).astype(np.int64), align_start=st.astype(np.float32),
                                   align_end=en.astype(np.float32))))
    fe = LogMelFbank(fs=fs, n_fft=2048, win_length=1200, hop_length=hop, n_mels=80, fmin=80, fmax=7600, device=dev)
    coll = MLMCollateFn(fe, mlm_prob=0.8, mean_phn_span=8, sega_emb=True, device_out=True)   # the whole batch dict is built on the device
    np.random.seed(1)
    coll(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        _, out = coll(data)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    frames = int(out["speech"].shape[0] * out["speech"].shape[1])
    res = dict(metric="collate mel-frames/s (host waveforms in, device batch dict out)", frames_per_s=frames / dt, ms=dt * 1e3,
               workload=f"B={B} utterances x {Tm} frames (12.5 s at 24 kHz), {Tp} phones each")
    try:
        from oracle import a3t_oracle as O
        np.random.seed(1)
        t0 = time.perf_counter()
        O.collate(data, O.A3TConfig())
        res["cpu_frames_per_s"] = frames / (time.perf_counter() - t0)
        res["cpu_kind"] = "port (oracle collate, torch.stft on host threads)"
    except Exception as e:  # noqa: BLE001
        res["cpu_frames_per_s"] = None
        res["cpu_note"] = type(e).__name__
    return res


def cpu_baseline_worker(blocks, Tm, Tp, threads, budget_s):
    """Runs in a child process: the oracle (CPU restatement of the reference) fwd+bwd+clip+Adam on
    `threads` host cores over a bounded sample of the same workload.  Prints one JSON line."""
    from oracle import a3t_oracle as O
    torch.set_num_threads(threads)
    oc = O.A3TConfig(enc_blocks=blocks, dec_blocks=blocks)
    B = 4
    batch = O.synthetic_batch(oc, B, Tm, Tp, seed=1234)
    p = O.to_torch_state(O.procedural_state(O.param_shapes(oc), 0), requires_grad=True)
    params = [t for t in p.values() if t.requires_grad]
    m = [torch.zeros_like(t) for t in params]
    v = [torch.zeros_like(t) for t in params]

    def one(step):
        for t in params:
            t.grad = None
        loss, _, _ = O.forward_loss(p, batch, oc, True)
        loss.backward()
        with torch.no_grad():
            O.clip_adam_step([t.data for t in params], [t.grad for t in params], m, v, step,
                             O.noam_lr(step, 1.0, oc.adim, 4000))

    t0 = time.time()
    one(1)
    warm = time.time() - t0
    n = max(1, min(3, int((budget_s - warm) / max(warm, 1e-3))))
    t0 = time.time()
    for i in range(n):
        one(2 + i)
    dt = (time.time() - t0) / n
    print(json.dumps(dict(value=B * Tm / dt, unit="mel-frames/s", cores=threads, kind="port",
                          sample=f"oracle fwd+bwd+clip+Adam fp32, {blocks}+{blocks} blocks, B={B} T_mel={Tm} "
                                 f"T_phn={Tp}, 1 warm-up + {n} timed steps ({dt:.2f} s/step)")), flush=True)


def cpu_baseline(blocks, Tm, Tp, budget_s=20.0, hard_timeout_s=150.0):
    """Bounded: child process + hard timeout, so the default bench run always finishes."""
    import subprocess
    threads = min(os.cpu_count() or 1, 32)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--blocks", str(blocks), "--tmel",
           str(Tm), "--tphn", str(Tp), "--threads", str(threads), "--budget", str(budget_s)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=hard_timeout_s,
                           env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # noqa: BLE001
        return dict(value=None, unit="mel-frames/s", cores=threads, kind="port",
                    sample=f"cpu baseline did not finish within {hard_timeout_s:.0f}s ({type(e).__name__})")


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _spawned_rank(local_rank, n, port, argv):
    """Entry of one self-launched rank (torch.multiprocessing.spawn): the same environment a launcher would set."""
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.argv = argv
    main()


def fake_cpu_rank(a, rank, world):
    """--fake-cpu: the launch / rendezvous / max-over-ranks / JSON plumbing of an N-rank run on the gloo backend with
    a trivial step (one SUM all-reduce of a small CPU tensor).  No kernels, no oracle: a CPU test of the N>1 launch path
    only (tests/test_distributed_cpu.py); its numbers mean nothing and the line says so."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.full((1024,), float(rank + 1))
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        dist.all_reduce(g.clone())
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    chk = torch.tensor([float(rank + 1)])
    dist.all_reduce(chk)
    if rank == 0:
        print(json.dumps({"metric": "mel-frames/sec (train)", "value": None, "unit": "mel-frames/s",
                          "n_gpus": world, "dist_world_size": dist.get_world_size(), "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": float(t) / max(a.steps, 1) * 1e3, "fake_cpu": True,
                          "rank_sum": float(chk), "data": "none (launch-path test, gloo, no kernels)"}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--compute", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tmel", type=int, default=1000)
    ap.add_argument("--tphn", type=int, default=120)
    ap.add_argument("--blocks", type=int, default=6, help="encoder blocks = decoder blocks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-dropout", action="store_true", help="disable the recipe's dropout sites (debug only)")
    ap.add_argument("--no-vocoder", action="store_true")
    ap.add_argument("--no-collate", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the BASELINE configs[3] leg (d=512, H=4, ff=2048, B=16, T_mel=1600)")
    ap.add_argument("--cpu-baseline-worker", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--fake-cpu", action="store_true", help="N-rank launch-path test on gloo/CPU (no kernels)")
    ap.add_argument("--one-gpu-gloo", action="store_true",
                    help="dry run of the N-rank path on a 1-GPU box: every rank on cuda:0, gloo as the transport (RCCL refuses two "
                         "ranks on one device); the line it prints is marked and is not a measurement")
    ap.add_argument("--comm-dtype", default="f32", choices=["f32", "bf16"], help="gradient all-reduce bucket dtype")
    ap.add_argument("--strict-traffic", action="store_true", help="exit instead of reporting traffic = null when this round's PMC file is missing")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with two rocprofv3 --pmc passes of a child process (the tracked file is used)")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--budget", type=float, default=20.0)
    a = ap.parse_args()
    if a.cpu_baseline_worker:
        cpu_baseline_worker(a.blocks, a.tmel, a.tphn, a.threads, a.budget)
        return
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline(a.blocks, a.tmel, a.tphn, a.budget)))
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks here (one process per GPU), rank 0 prints the JSON line
        import torch.multiprocessing as mp
        mp.spawn(_spawned_rank, args=(a.gpus, _free_port(), list(sys.argv)), nprocs=a.gpus, join=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.fake_cpu:
        fake_cpu_rank(a, rank, world)
        return
    if a.one_gpu_gloo:
        local = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.one_gpu_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but the RCCL group has {dist.get_world_size()} ranks")

    from a3t_amd import ops
    from a3t_amd.collate import synthetic_batch
    from a3t_amd.config import config_c2
    cfg = config_c2(enc_blocks=a.blocks, dec_blocks=a.blocks)
    B, Tm, Tp = a.batch, a.tmel, a.tphn
    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    log("building trainer")
    tr = build_trainer(cfg, dev, a.compute, world, dropout=not a.no_dropout,
                       comm_dtype=torch.bfloat16 if a.comm_dtype == "bf16" else torch.float32)
    batch = synthetic_batch(cfg, B, Tm, Tp, seed=1234 + rank, device=dev)
    log(f"params {tr.store.n_params}; warm-up")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.step(batch)
    sync()
    log("timed region")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.step(batch)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = dt / a.steps * 1e3
    frames = B * Tm * world
    value = frames / (ms * 1e-3)
    step_flops = 3.0 * fwd_flops_per_step(cfg, B, Tm, Tp)
    final_loss = float(loss)
    log(f"{ms:.1f} ms/step, {value:.0f} frames/s, loss {final_loss:.4f}")

    comm = None
    if world > 1:
        # the gradient exchange alone: the flat buffer in the trainer's buckets, 3 repetitions; bus bandwidth as
        # RCCL defines it for all-reduce: 2 (N-1)/N x bytes / time (what each xGMI link pair has to carry)
        gbuf = torch.zeros_like(tr.store.grad) if tr.comm_dtype == torch.float32 else \
            torch.zeros(tr.store.grad.numel(), dtype=tr.comm_dtype, device=dev)
        rngs = getattr(tr, "ranges", [(0, gbuf.numel())])
        for lo, hi in rngs:
            dist.all_reduce(gbuf[lo:hi])
        sync()
        t1 = time.perf_counter()
        for _ in range(3):
            for lo, hi in rngs:
                dist.all_reduce(gbuf[lo:hi])
        sync()
        tc = (time.perf_counter() - t1) / 3
        nbytes = gbuf.numel() * gbuf.element_size()
        comm = dict(backend=dist.get_backend(), dist_world_size=dist.get_world_size(), bytes_per_step=nbytes,
                    buckets=len(rngs), dtype=a.comm_dtype, allreduce_ms_alone=tc * 1e3,
                    algbw_GBps=nbytes / tc / 1e9, busbw_GBps=2.0 * (world - 1) / world * nbytes / tc / 1e9)
        del gbuf
        # how much of the exchange the backward pass hides: the same step with the collectives switched off (every rank on
        # its own; the replicas diverge from here on, nothing is timed after this) against the timed step
        saved = (tr.reducer, tr.world)
        tr.reducer, tr.world = None, 1
        for _ in range(2):
            tr.step(batch)
        sync()
        t2 = time.perf_counter()
        for _ in range(5):
            tr.step(batch)
        sync()
        t_nc = torch.tensor([(time.perf_counter() - t2) / 5], device=dev, dtype=torch.float64)
        dist.all_reduce(t_nc, op=dist.ReduceOp.MAX)
        tr.reducer, tr.world = saved
        ms_nc = float(t_nc) * 1e3
        exposed = max(0.0, ms - ms_nc)
        comm.update(step_ms_without_allreduce=ms_nc, allreduce_ms_exposed=exposed,
                    allreduce_ms_hidden=max(0.0, tc * 1e3 - exposed))
    roofline = None
    prof_rows = []
    if not a.no_kernel_profile:      # every rank runs the two extra steps (they contain collectives when world > 1)
        def gemm_profile():
            """One extra, untimed step with every GEMM launch bracketed by HIP events on the stream it is launched
            on; kernel names come from the library (a3t_gemm_last_kernel), as rocprofv3 prints them."""
            ops.PROFILE = []
            tr.step(batch)
            torch.cuda.synchronize()
            prof, ops.PROFILE = ops.PROFILE, None
            prof_rows[:] = prof
            agg = {}
            for name, flops, e0, e1, _shape in prof:
                s_ = agg.setdefault(name, [0.0, 0.0, 0])
                s_[0] += flops
                s_[1] += e0.elapsed_time(e1) * 1e-3
                s_[2] += 1
            return agg

        agg = gemm_profile()                       # as timed: weight-gradient GEMMs overlap on the side stream
        agg_rows = list(prof_rows)
        eng = tr.engine
        side, eng.side = eng.side, None            # same step, one stream: every kernel alone on the GPU
        alone = gemm_profile()
        alone_rows = list(prof_rows)
        prof_rows[:] = agg_rows
        eng.side = side
        # what the second stream buys: the same steps on one stream, timed like the timed region (5 steps)
        sync()
        side, eng.side = eng.side, None
        for _ in range(2):
            tr.step(batch)
        sync()
        t3 = time.perf_counter()
        for _ in range(5):
            tr.step(batch)
        sync()
        ms_one = (time.perf_counter() - t3) / 5 * 1e3
        eng.side = side
        side_ab = dict(ms_per_step_two_streams=ms, ms_per_step_one_stream=ms_one, gain_ms=ms_one - ms,
                       kernel_time_sum_in_step_ms=None)
        fwd_shapes = {(cfg.ff, cfg.ff_kernel * cfg.adim), (cfg.adim, cfg.ff_kernel * cfg.ff)}
        wg_shapes = {(cfg.ff, cfg.ff_kernel * cfg.adim), (cfg.adim, cfg.ff_kernel * cfg.ff)}
        sync()
    if rank == 0 and not a.no_kernel_profile:
        name, (fl, tt, n) = max(agg.items(), key=lambda kv: kv[1][1])
        achieved = fl / tt / 1e12
        peak = MFMA_PEAK["bf16" if "bf16" in name else "f32"]
        # HBM traffic of that kernel: ONLY from THIS round's PMC passes of this command (tools/r06_profiles.sh -> separate
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE x2 for gfx950); no silent fall-back to an older round's file
        traffic, traffic_source = None, None
        tfp = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if os.path.exists(tfp):
            for k, v in json.load(open(tfp)).items():
                if name in k:
                    traffic = v["hbm_bytes_per_launch"]
                    traffic_source = ("profiles/" + TRAFFIC_FILE + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      "command, FETCH_SIZE x2 for gfx950; a tracked file, not measured in this run)")
        traffic_tracked = traffic
        if world == 1 and not a.no_live_traffic:
            log(f"HBM traffic of {name}: two rocprofv3 --pmc passes of a child process")
            lt, lf, lw, lnote = live_hbm_traffic(name, a)
            if lt is not None:
                traffic = lt
                traffic_source = (f"measured in THIS run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes with --kernel-trace "
                                  f"over one step of this command in a child process; fetch {lf / 1e6:.1f} MB (FETCH_SIZE x2 for gfx950) + "
                                  f"write {lw / 1e6:.1f} MB per launch")
            elif traffic is not None:
                traffic_source += f"; the live measurement was not available ({lnote})"
        if traffic is None:
            traffic_source = (f"MISSING: profiles/{TRAFFIC_FILE} has no entry for {name} -- run tools/r06_profiles.sh on the GPU box and "
                              "commit its output; traffic is null, not borrowed from an older round")
            print("[bench] WARNING: " + traffic_source, file=sys.stderr, flush=True)
            if a.strict_traffic:
                raise SystemExit(traffic_source)

        # algorithmic bytes of one launch of that kernel = the SINGLE-PASS minimum: every operand read once (the taps of a conv
        # share their rows), the output written once (a K split adds nothing); averaged over its launches of the profiled step
        def alg_bytes(nm, M_, N_, K_, b_, tp_):
            tp_ = max(tp_, 1)
            if "<2," in nm or "_tn_" in nm or "_tn3_" in nm:     # token reduction: dy [K][M] + x [K][N / taps] in bf16, dW [M][N] fp32
                return b_ * (2.0 * K_ * M_ + 2.0 * K_ * (N_ // tp_) + 4.0 * M_ * N_)
            return b_ * (2.0 * M_ * (K_ // tp_) + 2.0 * N_ * K_ + 2.0 * M_ * N_)
        alg = [alg_bytes(nm, M_, N_, K_, b_, tp_) for nm, _f, _e0, _e1, (M_, N_, K_, b_, tp_, sk_) in prof_rows if nm == name]
        traffic_alg = sum(alg) / len(alg) if alg else None
        fa, ta, na = alone.get(name, (fl, tt, n))
        # per-class table of the conv-FFN GEMMs (75 % of the FLOPs), from the one-stream pass: forward / data gradient /
        # weight gradient; MFMA-busy from the tracked counter pass (profiles/r02_gemm_mfma_busy.json) when present
        classes = {}
        seen_wgrad = False          # launch order: every conv-FFN GEMM before the first weight gradient belongs to the forward pass
        for nm, fl_, e0_, e1_, (M_, N_, K_, b_, tp_, sk_) in alone_rows:
            if tp_ != cfg.ff_kernel or b_ != 1 or (N_, K_) not in fwd_shapes and (M_, N_) not in wg_shapes:
                continue
            is_wg = sk_ > 1 or "<2," in nm or "_tn_" in nm or "_tn3_" in nm
            seen_wgrad = seen_wgrad or is_wg
            cls = "wgrad" if is_wg else ("dgrad" if seen_wgrad else "fwd")
            c_ = classes.setdefault(cls, [0, 0.0, 0.0])
            c_[0] += 1
            c_[1] += e0_.elapsed_time(e1_) * 1e-3
            c_[2] += fl_
        gemm_classes = {k: dict(launches=v[0], avg_us=v[1] / v[0] * 1e6, tflops=v[2] / v[1] / 1e12, frac=v[2] / v[1] / 1e12 / peak)
                        for k, v in classes.items()}
        # the five kernels with the most time inside the step, as timed (two streams) and alone (same step, one stream): the
        # inflation of what shares the GPU with the side stream's weight gradients stays visible
        top5 = []
        for nm5, (f5, t5, n5) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:5]:
            fa5, ta5, na5 = alone.get(nm5, (f5, t5, n5))
            top5.append(dict(kernel=nm5, launches=n5, in_step_avg_us=t5 / n5 * 1e6, alone_avg_us=ta5 / na5 * 1e6,
                             in_step_total_ms=t5 * 1e3, alone_total_ms=ta5 * 1e3, alone_tflops=fa5 / ta5 / 1e12,
                             alone_frac=fa5 / ta5 / 1e12 / peak))
        roofline = dict(bound="mfma", kernel=name, launches=n, avg_us=tt / n * 1e6, achieved=achieved, peak=peak,
                        unit="TFLOP/s", frac=achieved / peak, traffic=traffic, traffic_source=traffic_source,
                        traffic_tracked_file=traffic_tracked,
                        traffic_algorithmic=traffic_alg, ffn_gemm_classes_alone=gemm_classes,
                        note="durations from HIP events inside the step, where the kernel shares the GPU with the other streams' "
                             "kernels (weight gradients, attention dV / dK); 'alone' = same step on one stream",
                        alone=dict(avg_us=ta / na * 1e6, achieved=fa / ta / 1e12, frac=fa / ta / 1e12 / peak),
                        all_gemms_alone=dict(tflops=sum(v[0] for v in alone.values()) / sum(v[1] for v in alone.values()) / 1e12,
                                             ms=sum(v[1] for v in alone.values()) * 1e3),
                        gemm_time_share=sum(v[1] for v in alone.values()) / (ms * 1e-3),
                        top5_in_step_vs_alone=top5, side_stream=side_ab,
                        step_tflops=step_flops / (ms * 1e-3) / 1e12)
    if rank == 0:
        out = {
            "metric": "mel-frames/sec (train)", "value": value, "unit": "mel-frames/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.compute, "data": "synthetic",
            "config": {"workload": f"VCTK A3T masked-mel train step: {a.blocks}+{a.blocks} Conformer blocks d=384 "
                                   f"H=2 ff=1536(k3), B={B}/GPU, T_mel={Tm}, T_phn={Tp}, postnet 5x256x5, "
                                   f"fwd+bwd+clip+Adam, dropout {'off' if a.no_dropout else '0.2/0.2/0.2 + postnet 0.5 (recipe)'}",
                       "global_batch": B * world, "parallelism": f"dp{world}", "params": tr.store.n_params,
                       "masked_fraction": float(batch["masked_position"].float().mean()),
                       "algorithmic_tflop_per_step": step_flops / 1e12, "final_loss": final_loss,
                       "hbm_allocated_gb": torch.cuda.max_memory_allocated(dev) / 1e9},
            "roofline": roofline,
        }
        if comm is not None:
            out["comm"] = comm
        if a.one_gpu_gloo:
            out["dry_run"] = f"{world} ranks share cuda:0 over gloo: the N-rank code path, NOT a throughput measurement"
        if world == 1 and not a.no_vocoder:
            log("vocoder leg (ParallelWaveGAN v1, 8 x 1000 frames)")
            out["vocoder"] = vocoder_rtf(dev, cpu=not a.no_cpu_baseline)
            inf = infill_leg(dev, cpu=not a.no_cpu_baseline)
            out["vocoder"]["infill"] = inf
            out["vocoder"]["pipeline_rtf"] = (inf["ms"] + out["vocoder"]["ms"]) * 1e-3 / out["vocoder"]["audio_seconds"]
        if world == 1 and not a.no_c4:
            log("configs[3] leg (d=512 H=4 ff=2048 + x-vector, B=16, T_mel=1600)")
            del tr, batch
            torch.cuda.empty_cache()
            out["c4"] = c4_leg(dev, a.compute)
        if world == 1 and not a.no_collate:
            log("collate leg (on-device log-mel)")
            out["collate"] = collate_leg(dev)
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (oracle on host cores, child process)")
            out["cpu_baseline"] = cpu_baseline(a.blocks, Tm, Tp, a.budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
