"""Generate the golden vectors in tests/golden/*.npz by running the REFERENCE
(richardbaihe/a3t at /root/reference) itself, in the build container only.

    python tests/golden/make_golden.py

The reference is imported through stub modules for its missing third-party
dependencies (typeguard, librosa, ... -- SURVEY §8c); ``librosa.filters.mel`` is
replaced by this repo's Slaney restatement (mel matrix parity is unpinned, see
oracle/a3t_oracle.py header).  Nothing of the reference is copied: the fixtures
hold inputs/seeds and the reference's numeric outputs only.  Weights are
procedural (oracle.procedural_state) so only outputs are stored.
"""
import argparse
import importlib.machinery
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def install_stubs():
    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return True

        def __getattr__(self, k):
            return _Any()

    def _stub(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []

        def _ga(k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Any()

        m.__getattr__ = _ga
        sys.modules[name] = m
        return m

    for n in ["typeguard", "editdistance", "torch_complex", "torch_complex.tensor", "humanfriendly",
              "librosa", "librosa.filters", "wandb", "h5py", "kaldiio", "soundfile", "g2p_en", "jaconv",
              "tacotron_cleaner", "tacotron_cleaner.cleaners", "torch.utils.tensorboard"]:
        _stub(n)
    from oracle.a3t_oracle import slaney_mel

    def mel(sr, n_fft, n_mels, fmin, fmax, htk=False):
        assert not htk
        return slaney_mel(sr, n_fft, n_mels, fmin, fmax)

    sys.modules["librosa"].filters = sys.modules["librosa.filters"]
    sys.modules["librosa.filters"].mel = mel
    sys.path.insert(0, REF)


def build_ref_model(c, vocab):
    import copy
    import yaml
    from argparse import Namespace
    from espnet2.tasks.mlm import MLMTask

    conf = yaml.safe_load(open(os.path.join(REF, "egs2/vctk/sedit/conf/fsp2_conformer.yaml")))
    enc = copy.deepcopy(conf["encoder_conf"])
    dec = copy.deepcopy(conf["decoder_conf"])
    mc = copy.deepcopy(conf["model_conf"])
    enc.update(attention_dim=c.adim, attention_heads=c.heads, linear_units=c.ff, num_blocks=c.enc_blocks,
               cnn_module_kernel=c.enc_kernel)
    dec.update(attention_dim=c.adim, attention_heads=c.heads, linear_units=c.ff, num_blocks=c.dec_blocks,
               cnn_module_kernel=c.dec_kernel)
    mc.update(postnet_layers=c.postnet_layers, postnet_chans=c.postnet_chans, postnet_filts=c.postnet_filts,
              mlm_prob=c.mlm_prob, mean_phn_span=c.mean_phn_span)
    fconf = dict(n_fft=c.n_fft, hop_length=c.hop_length, win_length=c.win_length, fs=c.fs, fmin=c.fmin,
                 fmax=c.fmax, n_mels=c.n_mels)
    args = Namespace(token_list=(["<blank>", "<unk>", "<space>"] + [f"t{i}" for i in range(vocab - 4)] + ["<sos/eos>"]), odim=c.odim, input_size=c.idim,
                     feats_extract="fbank", feats_extract_conf=fconf, normalize=None, normalize_conf={},
                     use_scaled_pos_enc=False, encoder=conf["encoder"], encoder_conf=enc,
                     decoder=conf["decoder"], decoder_conf=dec, model_conf=mc, init=conf["init"])
    model = MLMTask.build_model(args)
    cargs = Namespace(**vars(args))
    cargs.feats_extract = "fbank"
    cargs.feats_extract_conf = fconf
    collate = MLMTask.build_collate_fn(cargs, train=True)
    return model, collate


def load_procedural(model, c, seed):
    import torch
    from oracle.a3t_oracle import param_shapes, procedural_state

    shapes = param_shapes(c)
    sd = model.state_dict()
    assert set(sd.keys()) == set(shapes.keys()), (set(sd.keys()) ^ set(shapes.keys()))
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    state = procedural_state(shapes, seed)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return state


def gen_masks(out):
    """phones_masking / get_segment_pos / align->frames, bit-exact."""
    import torch
    from espnet2.train.collate_fn import phones_masking, get_segment_pos, random_spans_noise_mask

    cases = {}
    rs = np.random.RandomState(7)
    idx = 0
    for (B, P, T, prob, span) in [(1, 1, 40, 0.8, 8), (1, 2, 40, 0.8, 8), (2, 30, 300, 0.8, 8),
                                  (8, 120, 1000, 0.8, 8), (3, 17, 200, 0.5, 3), (2, 30, 300, 1.0, 8),
                                  (2, 30, 300, 0.8, 0)]:
        lens = rs.randint(max(P // 2, 1), P + 1, size=B)
        lens[0] = P
        tl = rs.randint(T // 2, T + 1, size=B)
        tl[0] = T
        a_s = np.zeros((B, P), np.int32)
        a_e = np.zeros((B, P), np.int32)
        for b in range(B):
            n, L = lens[b], tl[b]
            cuts = np.sort(rs.choice(np.arange(1, L), size=n - 1, replace=False)) if n > 1 else np.array([], np.int64)
            bd = np.concatenate([[0], cuts, [L]])
            a_s[b, :n], a_e[b, :n] = bd[:-1], bd[1:]
        smask = np.arange(T)[None] < tl[:, None]
        seed = 100 + idx
        np.random.seed(seed)
        xs = torch.zeros(B, T, 80)
        mp, _ = phones_masking(xs, torch.from_numpy(smask)[:, None], torch.from_numpy(a_s), torch.from_numpy(a_e),
                               torch.from_numpy(lens), prob, span, None)
        after = np.random.randint(0, 2 ** 31 - 1)
        sp, tp = get_segment_pos(xs, torch.zeros(B, P, dtype=torch.long), torch.from_numpy(a_s),
                                 torch.from_numpy(a_e), torch.from_numpy(lens), True)
        cases[f"c{idx}"] = dict(B=B, P=P, T=T, prob=prob, span=span, seed=seed, lens=lens, tl=tl, a_s=a_s, a_e=a_e,
                                masked=mp.numpy(), sp=sp.numpy(), tp=tp.numpy(), rng_after=after)
        idx += 1
    # span_boundary (inference) case
    B, T = 2, 50
    sb = np.array([[10, 20], [5, 45]], np.int64)
    xs = torch.zeros(B, T, 80)
    smask = np.arange(T)[None] < np.array([50, 40])[:, None]
    mp, _ = phones_masking(xs, torch.from_numpy(smask)[:, None], torch.zeros(B, 3, dtype=torch.int32),
                           torch.zeros(B, 3, dtype=torch.int32), torch.tensor([3, 3]), 0.8, 8, torch.from_numpy(sb))
    cases["sb"] = dict(T=T, sb=sb, tl=np.array([50, 40]), masked=mp.numpy())
    flat = {}
    for k, d in cases.items():
        for kk, v in d.items():
            flat[f"{k}.{kk}"] = np.asarray(v)
    # raw T5 helper
    np.random.seed(5)
    flat["rsnm.len37"] = random_spans_noise_mask(37, 0.8, 8)
    flat["rsnm.len2"] = random_spans_noise_mask(2, 0.8, 8)
    # align seconds -> frames (float32 floor), incl. boundary values
    sec = np.array([0.0, 0.0125, 0.012499999, 0.0250001, 1.0, 3.9999, 0.3625, 12.5, 7.0125, 0.1 + 0.2],
                   dtype=np.float32)
    fr = torch.floor(24000 * torch.from_numpy(sec) / 300).int().numpy()
    flat["align.sec"], flat["align.frames"] = sec, fr
    np.savez_compressed(os.path.join(out, "masks.npz"), **flat)
    print("masks.npz", len(flat))


def gen_logmel(out):
    import torch
    from espnet2.tts.feats_extract.log_mel_fbank import LogMelFbank

    fe = LogMelFbank(fs=24000, n_fft=2048, win_length=1200, hop_length=300, n_mels=80, fmin=80, fmax=7600)
    rs = np.random.RandomState(3)
    lens = np.array([4800, 3001, 2999], np.int64)
    N = int(lens.max())
    t = np.arange(N) / 24000.0
    wav = np.zeros((3, N), np.float32)
    for b in range(3):
        f0 = 200.0 * (b + 1)
        sig = 0.3 * np.sin(2 * np.pi * (f0 * t + 2000.0 * t * t)) + 0.01 * rs.standard_normal(N)
        wav[b, :lens[b]] = sig[:lens[b]].astype(np.float32)
    feats, flen = fe(torch.from_numpy(wav), torch.from_numpy(lens))
    np.savez_compressed(os.path.join(out, "logmel.npz"), wav=wav, lens=lens, feats=feats.numpy(),
                        feats_lengths=flen.numpy())
    print("logmel.npz", feats.shape, flen)


def run_model(model, batch):
    import torch
    keys = ["speech", "text", "masked_position", "speech_mask", "text_mask", "speech_segment_pos",
            "text_segment_pos"]
    b = {k: batch[k] for k in keys}
    loss, stats, weight = model(**b)
    return loss, weight


def gen_e2e(out):
    import torch
    from oracle.a3t_oracle import A3TConfig, tiny_config, synthetic_batch, param_shapes

    # ---- tiny config, train-mode BN, padded batch, full grads
    c = tiny_config()
    model, collate = build_ref_model(c, c.vocab)
    load_procedural(model, c, seed=1)
    batch = synthetic_batch(c, B=2, T_mel=48, T_phn=8, seed=11, lengths=[48, 37], text_lengths=[8, 6])
    model.train()
    before, after, _, _ = model._forward(dict(speech_pad=batch["speech"], text_pad=batch["text"],
                                              masked_position=batch["masked_position"],
                                              speech_mask=batch["speech_mask"], text_mask=batch["text_mask"],
                                              speech_segment_pos=batch["speech_segment_pos"],
                                              text_segment_pos=batch["text_segment_pos"]),
                                         batch["speech_segment_pos"])
    load_procedural(model, c, seed=1)  # reset BN running stats touched by the probe forward
    model.train()
    model.zero_grad()
    loss, weight = run_model(model, batch)
    loss.backward()
    d = dict(loss=loss.detach().numpy(), weight=weight.numpy(), before=before.detach().numpy(),
             after=after.detach().numpy())
    for n, p in model.named_parameters():
        d["grad." + n] = p.grad.numpy().copy()
    for n, b in model.named_buffers():
        if "running" in n or "num_batches" in n:
            d["buf." + n] = b.numpy().copy()
    # eval-mode BN
    load_procedural(model, c, seed=1)
    model.eval()
    with torch.no_grad():
        loss_e, _ = run_model(model, batch)
        out_inf = model.inference(**{k: batch[k][:1] for k in ["speech", "text", "masked_position", "speech_mask",
                                                                "text_mask", "speech_segment_pos", "text_segment_pos"]},
                                  span_boundary=[10, 30], use_teacher_forcing=True)
    d["loss_eval"] = loss_e.numpy()
    d["infer_splice"] = torch.cat([out_inf["feat_gen"][0][0], out_inf["feat_gen"][1], out_inf["feat_gen"][2][0]],
                                  dim=0).numpy()
    np.savez_compressed(os.path.join(out, "e2e_tiny.npz"), **d)
    print("e2e_tiny.npz loss", float(loss), "eval", float(loss_e))

    # ---- attention / conv-module / block level taps on a d=384,H=2 block (T=37 incl. fully masked utt)
    c1 = A3TConfig(enc_blocks=1, dec_blocks=1, postnet_layers=2, postnet_chans=16)
    model, _ = build_ref_model(c1, c1.vocab)
    load_procedural(model, c1, seed=2)
    model.train()
    rs = np.random.RandomState(21)
    B, T = 3, 37
    x = torch.from_numpy(rs.standard_normal((B, T, 384)).astype(np.float32))
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    mask[1, 0, 25:] = False
    mask[2, 0, :] = False
    layer = model.encoder.encoders[0]
    pos = model.encoder.speech_embed[4].pe[:, :T]
    x1 = x.clone().requires_grad_(True)
    att = layer.self_attn(x1, x1, x1, pos, mask)
    g = torch.from_numpy(rs.standard_normal((B, T, 384)).astype(np.float32))
    (att * g).sum().backward()
    d = dict(x=x.numpy(), mask=mask.numpy(), g=g.numpy(), attn_out=att.detach().numpy(),
             attn_probs=layer.self_attn.attn.detach().numpy(), attn_dx=x1.grad.numpy())
    for n, p in layer.self_attn.named_parameters():
        if n in ("pos_bias_u", "pos_bias_v", "linear_q.weight", "linear_pos.weight", "linear_out.bias"):
            d["attn_grad." + n] = p.grad.numpy().copy()
    model.zero_grad()
    x2 = x.clone().requires_grad_(True)
    (y2, _), _ = layer((x2, pos), mask)
    (y2 * g).sum().backward()
    d["block_out"] = y2.detach().numpy()
    d["block_dx"] = x2.grad.numpy()
    d["block_gradnorm"] = np.array([float(p.grad.norm()) for _, p in layer.named_parameters()], np.float64)
    d["block_gradnames"] = np.array([n for n, _ in layer.named_parameters()])
    # rel_shift pure data movement
    for Tt in (5, 37):
        bd = torch.from_numpy(rs.standard_normal((1, 2, Tt, Tt)).astype(np.float32))
        d[f"relshift_in{Tt}"] = bd.numpy()
        d[f"relshift_out{Tt}"] = layer.self_attn.rel_shift(bd).numpy()
    np.savez_compressed(os.path.join(out, "block384.npz"), **d)
    print("block384.npz")

    # ---- reference yaml verbatim (4+4, d=384), B=2, T_mel=200, T_phn=30: outputs only
    c2 = A3TConfig()
    model, _ = build_ref_model(c2, c2.vocab)
    n_params = sum(p.numel() for p in model.parameters())
    load_procedural(model, c2, seed=3)
    batch = synthetic_batch(c2, B=2, T_mel=200, T_phn=30, seed=12, lengths=[200, 163], text_lengths=[30, 24])
    model.train()
    model.zero_grad()
    loss, weight = run_model(model, batch)
    loss.backward()
    names = [n for n, _ in model.named_parameters()]
    gn = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()], np.float64)
    gs = np.array([float(p.grad.double().sum()) for _, p in model.named_parameters()], np.float64)
    load_procedural(model, c2, seed=3)
    model.train()
    with torch.no_grad():
        before, after, _, _ = model._forward(dict(speech_pad=batch["speech"], text_pad=batch["text"],
                                                  masked_position=batch["masked_position"],
                                                  speech_mask=batch["speech_mask"], text_mask=batch["text_mask"],
                                                  speech_segment_pos=batch["speech_segment_pos"],
                                                  text_segment_pos=batch["text_segment_pos"]),
                                             batch["speech_segment_pos"])
    np.savez_compressed(os.path.join(out, "e2e_refyaml.npz"), loss=loss.detach().numpy(), n_params=n_params,
                        before=before.numpy().astype(np.float32), after=after.numpy().astype(np.float32),
                        grad_names=np.array(names), grad_norm=gn, grad_sum=gs)
    print("e2e_refyaml.npz loss", float(loss), "params", n_params)

    # ---- full collate through the reference MLMCollateFn (waveform -> batch dict)
    c3 = tiny_config()
    _, collate = build_ref_model(c3, c3.vocab)
    rs = np.random.RandomState(31)
    data = []
    for i, n in enumerate([6000, 4500]):
        wav = (0.1 * rs.standard_normal(n)).astype(np.float32)
        P = 5 - i
        F_ = n // 300 + 1
        cuts = np.sort(rs.choice(np.arange(1, F_ - 1), size=P - 1, replace=False))
        bd = np.concatenate([[0], cuts, [F_ - 1]])
        st = (bd[:-1] * 300 / 24000 + 1e-4).astype(np.float32)
        en = (bd[1:] * 300 / 24000 + 1e-4).astype(np.float32)
        data.append((f"utt{i}", dict(speech=wav, text=rs.randint(2, 9, size=P).astype(np.int64),
                                     align_start=st, align_end=en)))
    np.random.seed(77)
    uids, b = collate(data)
    d = {f"out.{k}": v.numpy() for k, v in b.items()}
    for i, (u, dd) in enumerate(data):
        for k, v in dd.items():
            d[f"in{i}.{k}"] = v
    np.savez_compressed(os.path.join(out, "collate.npz"), **d)
    print("collate.npz", {k: tuple(v.shape) for k, v in b.items()})


def gen_pwg(out):
    import torch
    from espnet2.gan_tts.parallel_wavegan import ParallelWaveGANGenerator
    from oracle.a3t_oracle import PWGConfig, pwg_param_shapes, procedural_state

    cfg = PWGConfig()
    g = ParallelWaveGANGenerator(upsample_params={"upsample_scales": list(cfg.upsample_scales)})
    g.remove_weight_norm()
    sd = g.state_dict()
    shapes = pwg_param_shapes(cfg)
    assert set(sd.keys()) == set(shapes.keys()), set(sd.keys()) ^ set(shapes.keys())
    for k in sd:
        assert tuple(sd[k].shape) == tuple(shapes[k]), (k, sd[k].shape, shapes[k])
    state = procedural_state(shapes, seed=4)
    # keep activations O(1) through 30 blocks: the upsample smoothing kernels average
    for k in state:
        if "up_layers" in k:
            state[k] = np.abs(state[k]) / np.abs(state[k]).sum()
    g.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    g.eval()
    rs = np.random.RandomState(41)
    c = rs.standard_normal((20, 80)).astype(np.float32)
    z = rs.standard_normal((6000, 1)).astype(np.float32)
    taps = {}
    hooks = []
    for l in range(2):
        hooks.append(g.conv_layers[l].register_forward_hook(
            lambda m, i, o, l=l: taps.update({f"x{l}": o[0].detach().numpy()[..., :256].copy(),
                                        f"skip{l}": o[1].detach().numpy()[..., :256].copy()})))
    with torch.no_grad():
        wav = g.inference(torch.from_numpy(c), torch.from_numpy(z))
    np.savez_compressed(os.path.join(out, "pwg.npz"), c=c, z=z, wav=wav.numpy(), n_params=sum(
        p.numel() for p in g.parameters()), receptive=g.receptive_field_size, **taps)
    print("pwg.npz", wav.shape, float(wav.abs().mean()))


def fake_phone_duration(phns):
    """Stand-in for the FastSpeech2 duration predictor of the inference driver (seconds per phone); the sedit
    fixtures and tests share it."""
    out = []
    for ph in phns:
        h = sum(map(ord, ph))
        out.append(0.0 if h % 11 == 0 else (0.05 if ph == "sp" else 0.03 + 0.01 * (h % 7)))
    return out


def gen_sedit(out):
    """Span arithmetic of the speech-editing inference driver (espnet2/bin/sedit_inference.py): the reference's own
    get_phns_and_spans / prepare_features_with_duration / get_masked_mel_boundary / duration_adjust_factor, with its
    external programs (HTK aligner, phonemiser, FastSpeech2 duration predictor, librosa.load) replaced by synthetic
    stand-ins whose outputs are stored as the fixture's inputs."""
    import json
    import random
    import torch

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return True

        def __getattr__(self, k):
            return _Any()

    for n in ["matplotlib", "matplotlib.pylab", "parallel_wavegan", "parallel_wavegan.utils", "ipywidgets", "IPython",
              "IPython.display", "espnet2.tasks.tts", "espnet2.bin.align_english"]:
        m = types.ModuleType(n)
        m.__spec__ = importlib.machinery.ModuleSpec(n, None)
        m.__path__ = []

        def _ga(k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Any()

        m.__getattr__ = _ga
        sys.modules[n] = m
    import espnet2.bin.sedit_inference as S
    torch.use_deterministic_algorithms(False)     # (the driver switches it on at import)

    lex = {"THE": ["DH", "AH0"], "CAT": ["K", "AE1", "T"], "SAT": ["S", "AE1", "T"], "ON": ["AA1", "N"],
           "A": ["AH0"], "MAT": ["M", "AE1", "T"], "DOG": ["D", "AO1", "G"], "RAN": ["R", "AE1", "N"],
           "FAST": ["F", "AE1", "S", "T"], "HOME": ["HH", "OW1", "M"], "TODAY": ["T", "AH0", "D", "EY1"],
           "QUIETLY": ["K", "W", "AY1", "AH0", "T", "L", "IY0"], "BLUE": ["B", "L", "UW1"]}
    words = sorted(lex)
    rng = random.Random(7)
    fs, hop = 24000, 300
    cases = []
    wavs = {}
    mlm = types.SimpleNamespace(feats_extract=types.SimpleNamespace(fs=fs, hop_length=hop))
    tries = 0
    while len(cases) < 40 and tries < 400:
        tries += 1
        n_old = rng.randrange(3, 8)
        old_words = [rng.choice(words) for _ in range(n_old)]
        kind = rng.choice(["replace", "insert", "delete", "append", "first", "last", "mask", "mask_rec"])
        a = rng.randrange(1, n_old - 1) if n_old > 2 else 1
        b = rng.randrange(a + 1, n_old) if a + 1 < n_old else a + 1
        fresh = [rng.choice(words) for _ in range(rng.randrange(1, 4))]
        if kind == "replace":
            new_words = old_words[:a] + fresh + old_words[b:]
        elif kind == "insert":
            new_words = old_words[:a] + fresh + old_words[a:]
        elif kind == "delete":
            new_words = old_words[:a] + old_words[b:]
        elif kind == "append":
            new_words = old_words + fresh
        elif kind == "first":
            new_words = fresh + old_words[1:]
        elif kind == "last":
            new_words = old_words[:-1] + fresh
        else:
            new_words = old_words[:a] + ["[MASK]"] + old_words[b:]
        # aligner stand-in: optional silences at the ends and between words
        t, idx, times2, word2phns = 0.0, 0, [], {}
        seq = []
        if rng.random() < 0.5:
            seq.append("sp")
        for i, w in enumerate(old_words):
            seq.append(w)
            if i + 1 < n_old and rng.random() < 0.3:
                seq.append("sp")
        if rng.random() < 0.5:
            seq.append("sp")
        for w in seq:
            phs = ["sp"] if w == "sp" else lex[w]
            word2phns[f"{idx}_{w}"] = " ".join(phs)
            idx += 1
            for ph in phs:
                d = round(rng.uniform(0.03, 0.2), 4)
                times2.append([ph, round(t, 4), round(t + d, 4)])
                t = round(t + d, 4)
        new_phns, new_w2p = [], {}
        for i, w in enumerate(new_words):
            phs = [w] if w == "[MASK]" else lex[w]
            new_w2p[f"{i}_{w}"] = phs
            new_phns.extend(phs)
        old_str, new_str = " ".join(w.lower() for w in old_words), " ".join(
            w if w == "[MASK]" else w.lower() for w in new_words)
        wav = (np.arange(int(np.ceil(t * fs)) + rng.randrange(0, 900), dtype=np.float32) % 1000.0) / 1000.0 + 0.001
        S.alignment = lambda wav_path, txt, _t=times2, _w=word2phns: (_t, _w)
        S.words2phns_yuan = lambda line, _p=new_phns, _w=new_w2p: (list(_p), _w)
        S.get_fs2_model = lambda path: (None, None)
        S.duration_predict = lambda phns, fs_, hop_, m, pr, w, sid=None: fake_phone_duration(phns)
        S.librosa.load = lambda path, sr=None, _w=wav: (_w, sr)
        opts = dict(mask_reconstruct=(kind == "mask_rec"), duration_adjust=rng.random() < 0.7,
                    start_end_sp=rng.random() < 0.3)
        try:
            spans = S.get_phns_and_spans("x.wav", old_str, new_str)
            res = S.prepare_features_with_duration(mlm, old_str, new_str, "x.wav", "fs2.pth", **opts)
        except Exception as e:      # the driver itself rejects some edits (e.g. nothing left to anchor on)
            continue
        mfa_start, mfa_end, old_phns, new_phns_out, rep, add = spans
        new_wav, phns_o, ns, ne, ob, nb = res
        key = f"w{len(cases)}"
        wavs[key + ".in"] = wav
        wavs[key + ".out"] = np.asarray(new_wav)
        cases.append(dict(kind=kind, old_str=old_str, new_str=new_str, times2=times2, word2phns=word2phns,
                          new_phns=new_phns, new_word2phns=new_w2p, opts=opts, wav=key,
                          spans=dict(mfa_start=mfa_start, mfa_end=mfa_end, old_phns=old_phns, new_phns=new_phns_out,
                                     replaced=list(rep), added=list(add)),
                          plan=dict(phns=list(phns_o), mfa_start=[float(x) for x in ns], mfa_end=[float(x) for x in ne],
                                    old_span_boundary=[int(x) for x in ob], new_span_boundary=[int(x) for x in nb])))
    # the two leaf helpers on hand-picked boundary inputs
    leaf = dict(
        boundary=[dict(args=[[0.0, 0.5, 1.0], [0.5, 1.0, 1.5], fs, hop, [1, 2]]),
                  dict(args=[[0.0, 0.0125, 0.025], [0.0125, 0.025, 0.0375001], fs, hop, [0, 3]]),
                  dict(args=[[0.0, 0.5], [0.5, 1.2], fs, hop, [2, 2]])],
        factor=[dict(args=[[.3, .4, .5, .6, .7, .8, .9], [.3, .4, .4, .7, .7, .9, .9], list("abcdefg")]),
                dict(args=[[.3, .4, .5, .6], [.3, .4, .4, .7], list("abcd")]),
                dict(args=[[.3, .4, .5, .6, .7, .8, .9, 1.0], [.3, 0, .4, .7, .7, .9, .9, .5],
                           ["a", "b", "sp", "d", "e", "f", "g", "h"]])])
    for d in leaf["boundary"]:
        d["out"] = [int(x) for x in S.get_masked_mel_boundary(*d["args"])]
    for d in leaf["factor"]:
        d["out"] = float(S.duration_adjust_factor(*d["args"]))
    json.dump(dict(fs=fs, hop=hop, cases=cases, leaf=leaf), open(os.path.join(out, "sedit.json"), "w"))
    np.savez_compressed(os.path.join(out, "sedit_wav.npz"), **wavs)
    kinds = {}
    for c_ in cases:
        kinds[c_["kind"]] = kinds.get(c_["kind"], 0) + 1
    print("sedit.json", len(cases), "cases", kinds, "tries", tries)


def gen_average(out):
    """average_nbest_models (espnet2/main_funcs/average_nbest_models.py) on four small epoch files."""
    import tempfile
    from pathlib import Path
    import torch
    from espnet2.main_funcs.average_nbest_models import average_nbest_models
    from espnet2.train.reporter import Reporter

    rs = np.random.RandomState(3)
    d = {}
    losses = {1: 0.9, 2: 0.4, 3: 0.7, 4: 0.5, 5: 0.45}
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        for e in losses:
            st = {"a.weight": torch.from_numpy(rs.standard_normal((3, 4)).astype(np.float32)),
                  "bn.running_var": torch.from_numpy(rs.rand(5).astype(np.float32)),
                  "bn.num_batches_tracked": torch.tensor(100 * e, dtype=torch.long)}
            torch.save(st, td / f"{e}epoch.pth")
            for k, v in st.items():
                d[f"in.{e}.{k}"] = v.numpy()
        rep = Reporter()
        rep.epoch = 5
        rep.stats = {e: {"valid": {"loss": v}, "train": {"loss": v * 2}} for e, v in losses.items()}
        average_nbest_models(td, rep, [("valid", "loss", "min")], [1, 3, 9])
        files = sorted(p.name for p in td.iterdir())
        links = {p.name: os.readlink(p) for p in td.iterdir() if p.is_symlink()}
        ave = torch.load(td / "valid.loss.ave_3best.pth")
        for k, v in ave.items():
            d["out.ave3." + k] = v.numpy()
    d["losses"] = np.array([[e, v] for e, v in losses.items()])
    d["files"] = np.array(files)
    d["links"] = np.array([f"{k}->{v}" for k, v in sorted(links.items())])
    np.savez_compressed(os.path.join(out, "average.npz"), **d)
    print("average.npz", files, links)


def grad_sample_index(name, numel, n=256):
    """Deterministic sample of flat indices of one parameter (shared with tests/test_gpu_parity_r2.py)."""
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return rs.randint(0, numel, size=min(n, numel))


def _fwd_parts(model, batch):
    return model._forward(dict(speech_pad=batch["speech"], text_pad=batch["text"],
                               masked_position=batch["masked_position"], speech_mask=batch["speech_mask"],
                               text_mask=batch["text_mask"], speech_segment_pos=batch["speech_segment_pos"],
                               text_segment_pos=batch["text_segment_pos"]), batch["speech_segment_pos"])


def gen_extra(out):
    """Round-2 fixtures (e2e_extra.npz): a model WITHOUT a postnet (after_outs is None, sedit_model.py:369-374),
    BASELINE configs[0] exactly (C1: 1+1 blocks, d=128, H=2, ff=512, postnet 5x256x5, B=2, T_mel=200, T_phn=30), the
    d=512 / H=4 shape family of configs[3] at a size the CPU runs, and SAMPLED elements of every parameter gradient of
    the reference-yaml (4+4, d=384) step whose e2e_refyaml.npz only keeps norms (full gradients would be 270 MB)."""
    import torch
    from oracle.a3t_oracle import A3TConfig, tiny_config, synthetic_batch
    d = {}

    def run(tag, c, seed, batch, full_grads):
        model, _ = build_ref_model(c, c.vocab)
        load_procedural(model, c, seed=seed)
        model.train()
        with torch.no_grad():
            before, after, _, _ = _fwd_parts(model, batch)
        load_procedural(model, c, seed=seed)
        model.train()
        model.zero_grad()
        loss, _ = run_model(model, batch)
        loss.backward()
        d[tag + ".loss"] = loss.detach().numpy()
        d[tag + ".before"] = before.numpy().astype(np.float32)
        if after is not None:
            d[tag + ".after"] = after.numpy().astype(np.float32)
        names = [n for n, _ in model.named_parameters()]
        d[tag + ".grad_names"] = np.array(names)
        d[tag + ".grad_norm"] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
        for n, p in model.named_parameters():
            g = p.grad.numpy().reshape(-1)
            if full_grads:
                d[f"{tag}.grad.{n}"] = p.grad.numpy().copy()
            else:
                d[f"{tag}.gsample.{n}"] = g[grad_sample_index(n, g.size)].copy()
        print(tag, "loss", float(loss), "params", sum(p.numel() for p in model.parameters()))

    c0 = tiny_config(postnet_layers=0, postnet_chans=0, postnet_filts=0)
    run("nopost", c0, 1, synthetic_batch(c0, B=2, T_mel=48, T_phn=8, seed=11, lengths=[48, 37], text_lengths=[8, 6]), True)
    c1 = A3TConfig(adim=128, heads=2, ff=512, enc_blocks=1, dec_blocks=1)
    run("c1", c1, 4, synthetic_batch(c1, B=2, T_mel=200, T_phn=30, seed=13, lengths=[200, 171], text_lengths=[30, 22]), False)
    c4 = A3TConfig(adim=512, heads=4, ff=2048, enc_blocks=1, dec_blocks=1)
    run("c4s", c4, 5, synthetic_batch(c4, B=2, T_mel=96, T_phn=16, seed=14, lengths=[96, 70], text_lengths=[16, 11]), False)
    c2 = A3TConfig()
    run("refyaml", c2, 3, synthetic_batch(c2, B=2, T_mel=200, T_phn=30, seed=12, lengths=[200, 163], text_lengths=[30, 24]), False)
    for k in [k for k in d if k.startswith("refyaml.") and not k.startswith("refyaml.gsample.")]:
        del d[k]          # loss / outputs / norms of this step are already in e2e_refyaml.npz
    np.savez_compressed(os.path.join(out, "e2e_extra.npz"), **d)
    print("e2e_extra.npz", len(d), "arrays")

    # ---- alignment dtype (collate_fn.py:236-237): the collate keeps the dtype of the alignment arrays it is handed --
    # float32 from the dataset, float64 from sedit_inference.py:603-604 (np.array of Python floats) -- and
    # floor(fs * t / hop) differs between the two for boundaries a hair below a frame edge.  Same utterance, both dtypes.
    c3 = tiny_config()
    _, collate = build_ref_model(c3, c3.vocab)
    rs = np.random.RandomState(41)
    n = 9000
    wav = (0.1 * rs.standard_normal(n)).astype(np.float32)
    F_ = n // 300 + 1
    cuts = np.sort(rs.choice(np.arange(2, F_ - 1), size=5, replace=False))
    bd = np.concatenate([[0], cuts, [F_ - 1]]).astype(np.float64)
    st64, en64 = bd[:-1] * 300 / 24000 - 1e-8, bd[1:] * 300 / 24000 - 1e-8
    st64[0] = 0.0
    text = rs.randint(2, 9, size=6).astype(np.int64)
    e = {}
    for tag, cast in (("f64", np.float64), ("f32", np.float32)):
        np.random.seed(5)
        _, b = collate([("u", dict(speech=wav, text=text, align_start=st64.astype(cast), align_end=en64.astype(cast)))])
        for k in ("speech_segment_pos", "text_segment_pos", "masked_position"):
            e[f"{tag}.{k}"] = b[k].numpy()
    assert not np.array_equal(e["f64.speech_segment_pos"], e["f32.speech_segment_pos"])
    e.update(wav=wav, text=text, align_start=st64, align_end=en64)
    np.savez_compressed(os.path.join(out, "align_dtype.npz"), **e)
    print("align_dtype.npz: segment ids differ at", int((e["f64.speech_segment_pos"] != e["f32.speech_segment_pos"]).sum()), "frames")


def gen_bf16ref(out):
    """e2e_bf16ref.npz: what the REFERENCE ITSELF does to its mel outputs when its matrix products run in bf16
    (torch.autocast("cpu", dtype=torch.bfloat16): Linear / Conv1d / matmul operands and results in bf16, LayerNorm / softmax
    / BatchNorm statistics in fp32) -- the yardstick the bf16 tolerances of tests/test_gpu_parity_r2.py are pinned to, instead
    of a self-granted waiver.  Same configurations, procedural weights and batches as e2e_extra.npz; the batch is padded
    to the 16-byte DMA granule exactly as the product's plugin model pads it (so both see the same tensors).  Stored: the
    reference's bf16 outputs and their max / RMS error against its own fp32 outputs, relative to max(1, max|fp32|)."""
    import torch
    from oracle.a3t_oracle import A3TConfig, synthetic_batch
    from a3t_amd.espnet_model import ESPnetMLMEncAsDecoderModel
    # The reference cannot take bf16 scores as it stands: attention.py:81 asks numpy for finfo of the score dtype and numpy has
    # no bfloat16 ("Got unsupported ScalarType BFloat16").  The one change made here (monkeypatch, nothing is copied): the
    # scores enter forward_attention promoted to fp32 -- bf16 logits, fp32 softmax, the split a3t_amd's bf16 mode uses too.
    from espnet.nets.pytorch_backend.transformer import attention as ref_attention
    _orig_fa = ref_attention.MultiHeadedAttention.forward_attention
    ref_attention.MultiHeadedAttention.forward_attention = lambda self, value, scores, mask: _orig_fa(self, value, scores.float(), mask)
    d = {}
    cases = dict(c1=(A3TConfig(adim=128, heads=2, ff=512, enc_blocks=1, dec_blocks=1), 4,
                     dict(B=2, T_mel=200, T_phn=30, seed=13, lengths=[200, 171], text_lengths=[30, 22])),
                 c4s=(A3TConfig(adim=512, heads=4, ff=2048, enc_blocks=1, dec_blocks=1), 5,
                      dict(B=2, T_mel=96, T_phn=16, seed=14, lengths=[96, 70], text_lengths=[16, 11])),
                 refyaml=(A3TConfig(), 3, dict(B=2, T_mel=200, T_phn=30, seed=12, lengths=[200, 163], text_lengths=[30, 24])))
    for tag, (c, seed, bk) in cases.items():
        batch = ESPnetMLMEncAsDecoderModel._pad_to_dma_granule(dict(synthetic_batch(c, **bk)))
        model, _ = build_ref_model(c, c.vocab)
        load_procedural(model, c, seed=seed)
        model.train()
        with torch.no_grad():
            b32, a32, _, _ = _fwd_parts(model, batch)
        load_procedural(model, c, seed=seed)          # (train-mode BatchNorm moved the running statistics)
        model.train()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            b16, a16, _, _ = _fwd_parts(model, batch)
        for name, x32, x16 in (("before", b32, b16), ("after", a32, a16)):
            x32, x16 = x32.double().numpy(), x16.float().double().numpy()
            scale = max(1.0, float(np.abs(x32).max()))
            d[f"{tag}.{name}.bf16"] = x16.astype(np.float32)
            d[f"{tag}.{name}.err_max"] = np.float64(np.abs(x16 - x32).max() / scale)
            d[f"{tag}.{name}.err_rms"] = np.float64(np.sqrt(np.mean((x16 - x32) ** 2)) / scale)
            print(f"{tag} {name}: reference under bf16 autocast vs its own fp32: max {d[f'{tag}.{name}.err_max']:.3e}"
                  f" rms {d[f'{tag}.{name}.err_rms']:.3e} of scale {scale:.2f}")
    ref_attention.MultiHeadedAttention.forward_attention = _orig_fa
    np.savez_compressed(os.path.join(out, "e2e_bf16ref.npz"), **d)
    print("e2e_bf16ref.npz", len(d), "arrays")


def sweep(out, n_masks, n_models):
    """Randomised pinning of the ORACLE to the REFERENCE (container only; the fixed goldens above are what travels).

    * index work: random batch shapes / alignments / mask settings -> the reference's collate_fn.phones_masking and
      get_segment_pos vs the oracle's, bit-exact, plus the numpy global RNG state left behind;
    * numerics: random small architectures (width, heads, FFN, block counts, conv-module kernels, postnet) and ragged
      batches -> the reference model's loss, outputs and EVERY parameter gradient vs the oracle's (fp32, dropout 0).
    Writes sweep_report.json (counts, worst errors, seed) next to the fixtures."""
    import json
    import random
    import torch
    from espnet2.train.collate_fn import phones_masking, get_segment_pos, random_spans_noise_mask
    from oracle import a3t_oracle as O

    rng = random.Random(20260928)
    rep = dict(seed=20260928, mask_cases=0, mask_mismatches=0, rsnm_cases=0, model_cases=0, model_mismatches=0,
               worst_loss_rel=0.0, worst_grad_rel=0.0)
    for case in range(n_masks):
        B = rng.randrange(1, 5)
        T = rng.randrange(4, 200)
        P = rng.randrange(1, min(30, T - 1) + 1)
        a_s = np.zeros((B, P), np.int32)
        a_e = np.zeros((B, P), np.int32)
        lens, nonpad = [], np.zeros((B, T), bool)
        for b in range(B):
            L = rng.randrange(max(2, P), T + 1) if b else T
            pb = rng.randrange(1, P + 1) if b else P
            cuts = sorted(rng.sample(range(1, L), pb - 1)) if pb > 1 else []
            bd = [0] + cuts + [L]
            a_s[b, :pb], a_e[b, :pb] = bd[:-1], bd[1:]
            lens.append(pb)
            nonpad[b, :L] = True
        prob, span = rng.choice([(0.15, 3), (0.15, 8), (0.5, 3), (0.5, 8), (0.8, 8), (1.0, 3), (1.0, 0), (0.8, 0),
                                 (0.5, 0)])
        if span == 0 and T < 24:
            span = 8
        sb = None
        if rng.random() < 0.2:
            s0 = rng.randrange(0, T - 1)
            sb = np.array([[s0, rng.randrange(s0 + 1, T + 1)]] * B)
        seed = rng.randrange(1 << 30)
        xs = torch.zeros(B, T, 80)
        np.random.seed(seed)
        mp, _ = phones_masking(xs, torch.from_numpy(nonpad)[:, None], torch.from_numpy(a_s), torch.from_numpy(a_e),
                               torch.tensor(lens), prob, span, None if sb is None else torch.from_numpy(sb))
        st_ref = np.random.get_state()[1].copy()
        sp, tp = get_segment_pos(xs, torch.zeros(B, P, dtype=torch.long), torch.from_numpy(a_s),
                                 torch.from_numpy(a_e), torch.tensor(lens), True)
        np.random.seed(seed)
        got = O.phones_masking(T, nonpad, a_s, a_e, lens, prob, span, sb)
        st_got = np.random.get_state()[1].copy()
        sp_o, tp_o = O.get_segment_pos(T, P, a_s, a_e, lens, True)
        ok = (np.array_equal(np.asarray(got), mp.numpy()) and np.array_equal(st_ref, st_got)
              and np.array_equal(np.asarray(sp_o), sp.numpy()) and np.array_equal(np.asarray(tp_o), tp.numpy()))
        rep["mask_cases"] += 1
        if not ok:
            rep["mask_mismatches"] += 1
            print("MASK MISMATCH", case, B, T, P, prob, span, sb)
        Ln = rng.randrange(2, 400)
        pr2, sp2 = (prob, span) if (prob < 1.0 and span > 0) else (0.8, 8)
        np.random.seed(seed)
        r = random_spans_noise_mask(Ln, pr2, sp2)
        np.random.seed(seed)
        o = O.random_spans_noise_mask(Ln, pr2, sp2)
        rep["rsnm_cases"] += 1
        if not np.array_equal(np.asarray(r), np.asarray(o)):
            rep["mask_mismatches"] += 1
            print("RSNM MISMATCH", case, Ln, pr2, sp2)
    for case in range(n_models):
        heads = rng.choice([1, 2, 4])
        adim = heads * rng.choice([8, 16, 24])
        oc = O.A3TConfig(adim=adim, heads=heads, ff=rng.choice([24, 48, 64]), enc_blocks=rng.choice([1, 2]),
                         dec_blocks=rng.choice([1, 2]), enc_kernel=rng.choice([3, 7, 15]),
                         dec_kernel=rng.choice([7, 31]), postnet_layers=rng.choice([2, 3, 5]),
                         postnet_chans=rng.choice([16, 24]), vocab=rng.randrange(8, 40))
        B = rng.randrange(1, 4)
        T_mel, T_phn = rng.randrange(24, 90), rng.randrange(3, 12)
        lengths = [T_mel] + [rng.randrange(max(T_phn + 2, T_mel // 2), T_mel + 1) for _ in range(B - 1)]
        tlens = [T_phn] + [rng.randrange(2, T_phn + 1) for _ in range(B - 1)]
        seed = rng.randrange(1 << 20)
        batch = O.synthetic_batch(oc, B, T_mel, T_phn, seed=seed, lengths=lengths, text_lengths=tlens)
        model, _ = build_ref_model(oc, oc.vocab)
        state = load_procedural(model, oc, seed % 97)
        train_bn = rng.random() < 0.8
        model.train() if train_bn else model.eval()
        model.zero_grad()
        loss_r, _ = run_model(model, batch)
        loss_r.backward()
        p = O.to_torch_state(state, requires_grad=True)
        loss_o, before, after = O.forward_loss(p, batch, oc, train_bn)
        loss_o.backward()
        lrel = abs(float(loss_o.detach()) - float(loss_r.detach())) / max(abs(float(loss_r.detach())), 1e-12)
        worst = 0.0
        # gradients that are mathematically zero (a conv bias feeding train-mode BatchNorm) are rounding noise on
        # both sides: judge them against the model's largest gradient norm instead of their own
        gmax = max(float(q.grad.norm()) for q in model.parameters() if q.grad is not None)
        for n, q in model.named_parameters():
            g_o = p[n].grad
            if g_o is None:
                g_o = torch.zeros_like(q)
            g_r = q.grad if q.grad is not None else torch.zeros_like(q)
            den = max(float(g_r.norm()), 1e-3 * gmax)
            e = float((g_o - g_r).norm()) / den
            if e > worst:
                worst, worst_name = e, (n, float(g_r.norm()), float(g_o.norm()))
        rep["model_cases"] += 1
        rep["worst_loss_rel"] = max(rep["worst_loss_rel"], lrel)
        rep["worst_grad_rel"] = max(rep["worst_grad_rel"], worst)
        if lrel > 1e-5 or worst > 2e-4:
            rep["model_mismatches"] += 1
            print("MODEL MISMATCH", case, vars(oc), B, lengths, tlens, lrel, worst, worst_name)
    rep["criteria"] = dict(index="bit-exact incl. numpy RNG state", loss_rel=1e-5, grad_rel_l2=2e-4)
    json.dump(rep, open(os.path.join(out, "sweep_report.json"), "w"), indent=1)
    print("sweep_report.json", rep)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--sweep", type=int, nargs=2, default=None, metavar=("N_MASKS", "N_MODELS"),
                    help="randomised oracle-vs-reference sweep instead of regenerating the fixtures")
    a = ap.parse_args()
    install_stubs()
    import torch

    torch.manual_seed(0)
    torch.set_num_threads(8)
    if a.sweep:
        sweep(HERE, *a.sweep)
        sys.exit(0)
    todo = dict(masks=gen_masks, logmel=gen_logmel, e2e=gen_e2e, pwg=gen_pwg, sedit=gen_sedit, average=gen_average,
                extra=gen_extra, bf16ref=gen_bf16ref)
    for k, f in todo.items():
        if a.only and k not in a.only.split(","):
            continue
        f(HERE)
