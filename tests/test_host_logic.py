"""CPU tests of the product's host-side logic (collate mirror, init, config translation)."""
import os

import pytest

import numpy as np
import torch

from a3t_amd import collate as C
from a3t_amd.config import A3TConfig

G = os.path.join(os.path.dirname(__file__), "golden")


def test_host_masks_bit_exact_vs_reference_golden():
    g = np.load(os.path.join(G, "masks.npz"))
    for i in range(7):
        k = f"c{i}."
        P, T = int(g[k + "P"]), int(g[k + "T"])
        smask = np.arange(T)[None] < g[k + "tl"][:, None]
        np.random.seed(int(g[k + "seed"]))
        mp = C.phones_masking(T, smask, g[k + "a_s"], g[k + "a_e"], g[k + "lens"], float(g[k + "prob"]),
                              int(g[k + "span"]))
        assert np.array_equal(mp, g[k + "masked"]), i
        assert np.random.randint(0, 2 ** 31 - 1) == int(g[k + "rng_after"]), i
        sp, tp = C.get_segment_pos(T, P, g[k + "a_s"], g[k + "a_e"], g[k + "lens"], True)
        assert np.array_equal(sp, g[k + "sp"]) and np.array_equal(tp, g[k + "tp"])
    T = int(g["sb.T"])
    smask = np.arange(T)[None] < g["sb.tl"][:, None]
    mp = C.phones_masking(T, smask, np.zeros((2, 3), np.int32), np.zeros((2, 3), np.int32), [3, 3], 0.8, 8,
                          span_boundary=g["sb.sb"])
    assert np.array_equal(mp, g["sb.masked"])
    np.random.seed(5)
    assert np.array_equal(C.random_spans_noise_mask(37, 0.8, 8), g["rsnm.len37"])
    assert np.array_equal(C.random_spans_noise_mask(2, 0.8, 8), g["rsnm.len2"])
    assert np.array_equal(C.align_to_frames(g["align.sec"], 24000, 300), g["align.frames"])


def test_synthetic_batch_shapes_and_mask_fraction():
    c = A3TConfig()
    b = C.synthetic_batch(c, 4, 1000, 120, seed=1234)
    assert b["speech"].shape == (4, 1000, 80) and b["text"].shape == (4, 120)
    frac = float(b["masked_position"].float().mean())
    assert 0.7 < frac < 0.9
    assert int(b["speech_segment_pos"].max()) == 120 and int(b["speech_segment_pos"].min()) == 1


def test_init_matches_recipe_rules():
    from a3t_amd.init import xavier_init_
    from a3t_amd.params import ParamStore
    c = A3TConfig(adim=32, heads=2, ff=64, enc_blocks=1, dec_blocks=1, postnet_layers=2, postnet_chans=16, vocab=11)
    s = xavier_init_(ParamStore(c, "cpu"), seed=0)
    sd = s.state_dict()
    assert float(sd["encoder.encoders.0.conv_module.norm.weight"].abs().max()) == 0.0      # BN gamma zeroed
    assert float(sd["encoder.encoders.0.norm_mha.weight"].min()) == 1.0                     # LN reset
    assert float(sd["encoder.text_embed.0.weight"][-1].abs().max()) == 0.0                  # padding row
    w = sd["encoder.encoders.0.feed_forward.w_1.weight"]                                    # (ff, d, 3)
    bound = (6.0 / (32 * 3 + 64 * 3)) ** 0.5
    assert float(w.abs().max()) <= bound and float(w.abs().max()) > 0.8 * bound
    assert float(sd["sfc.bias"].abs().max()) == 0.0


def test_config_from_espnet_yaml_dicts():
    enc = dict(input_layer="sega_mlm", attention_dim=384, attention_heads=2, linear_units=1536, num_blocks=4,
               cnn_module_kernel=7, positionwise_layer_type="conv1d", positionwise_conv_kernel_size=3,
               macaron_style=True, use_cnn_module=True, dropout_rate=0.2)
    dec = dict(attention_dim=384, attention_heads=2, linear_units=1536, num_blocks=4, cnn_module_kernel=31)
    mc = dict(postnet_layers=5, postnet_chans=256, postnet_filts=5, lsm_weight=0.1, mlm_prob=0.8, mean_phn_span=8)
    c = A3TConfig.from_espnet(enc, dec, mc, input_size=80, odim=80, vocab=73)
    assert (c.enc_kernel, c.dec_kernel, c.ff, c.dk) == (7, 31, 1536, 192)


def test_host_index_logic_randomised_against_oracle():
    """300 random alignments / batch shapes / mask settings: the product's vectorised host index work
    (a3t_amd.collate: align_to_frames, random_spans_noise_mask, phones_masking, get_segment_pos) must be BIT-EXACT with
    the oracle's loop-level restatement of collate_fn.py under the same numpy RNG state, and leave the RNG in the same
    state (the global numpy RNG order is part of the contract)."""
    import random
    import torch
    from a3t_amd import collate as C
    from oracle import a3t_oracle as O
    rng = random.Random(0)
    for case in range(300):
        B = rng.randrange(1, 5)
        T_mel = rng.randrange(4, 200)
        P = rng.randrange(1, min(30, T_mel - 1) + 1)
        a_s = np.zeros((B, P), dtype=np.int32)
        a_e = np.zeros((B, P), dtype=np.int32)
        lens, nonpad = [], np.zeros((B, T_mel), dtype=bool)
        for b in range(B):
            L = rng.randrange(max(2, P), T_mel + 1) if b else T_mel
            pb = rng.randrange(1, P + 1) if b else P
            cuts = sorted(rng.sample(range(1, L), pb - 1)) if pb > 1 else []
            bounds = [0] + cuts + [L]
            a_s[b, :pb], a_e[b, :pb] = bounds[:-1], bounds[1:]
            lens.append(pb)
            nonpad[b, :L] = True
        # (prob, mean span) pairs inside the domain of the reference's T5 segmentation helper: it needs
        #  n_spans <= min(n_noise, n_non_noise) and raises otherwise, e.g. for prob 0.8 with span 3)
        prob, span = rng.choice([(0.15, 3), (0.15, 8), (0.5, 3), (0.5, 8), (0.8, 8), (1.0, 3), (1.0, 0), (0.8, 0), (0.5, 0)])
        if span == 0 and T_mel < 24:
            span = 8
        sb = None
        if rng.random() < 0.2:
            s0 = rng.randrange(0, T_mel - 1)
            sb = np.array([[s0, rng.randrange(s0 + 1, T_mel + 1)]] * B)
        seed = rng.randrange(1 << 30)
        np.random.seed(seed)
        ref = O.phones_masking(T_mel, nonpad, a_s, a_e, lens, prob, span, sb)
        st_ref = np.random.get_state()[1].copy()
        np.random.seed(seed)
        got = C.phones_masking(T_mel, nonpad, a_s, a_e, lens, prob, span, sb)
        st_got = np.random.get_state()[1].copy()
        assert np.array_equal(got, ref), (case, B, T_mel, P, prob, span)
        assert np.array_equal(st_got, st_ref), case
        sp_r, tp_r = O.get_segment_pos(T_mel, P, a_s, a_e, lens, True)
        sp_g, tp_g = C.get_segment_pos(T_mel, P, a_s, a_e, lens, True)
        assert np.array_equal(sp_g, sp_r) and np.array_equal(tp_g, tp_r), case
        # seconds -> frames, incl. stamps sitting exactly on frame boundaries
        sec = np.array([rng.choice([k * 300 / 24000, k * 300 / 24000 + 1e-4, rng.random() * 12.5]) for k in range(1, 40)],
                       dtype=np.float32)
        assert np.array_equal(C.align_to_frames(sec, 24000, 300), O.align_to_frames(torch.from_numpy(sec), 24000, 300).numpy())
        Ln = rng.randrange(2, 400)
        np.random.seed(seed)
        pr2, sp2 = (prob, span) if (prob < 1.0 and span > 0) else (0.8, 8)
        m_r = O.random_spans_noise_mask(Ln, pr2, sp2)
        np.random.seed(seed)
        m_g = C.random_spans_noise_mask(Ln, pr2, sp2)
        assert np.array_equal(m_g, m_r), (case, Ln)


def test_row_kernels_that_run_beside_the_weight_gradient_gemm_keep_their_register_footprint():
    """Inside the training step the LayerNorm backward (main stream) runs beside the weight-gradient GEMM of the side stream, whose two
    176-register waves per SIMD leave 160 of the 512 registers: at <= 80 registers two of its waves fit, at 132 (the half-wave layout it
    replaced at D = 384) one -- 93 against 80 us per launch inside the step, 63 launches (profiles/r05_ln_bwd_row64_ab.txt).  Held here
    against the compiler's own report, like the GEMM budgets below."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "a3t_amd", "lib", "norm_reduce.resources.json")
    if not os.path.exists(path):
        pytest.skip("library not built by a3t_amd.build in this tree")
    rows = json.load(open(path))
    hit = {n: r for n, r in rows.items() if n.startswith("_Z19ln_bwd_row64_kernelILi3E")}
    assert len(hit) == 2            # bf16 / fp32 incoming gradient
    for name, r in hit.items():
        assert r["VGPRs"] <= 80 and r["ScratchSize [bytes/lane]"] == 0 and r["LDS Size [bytes/block]"] <= 16384, (name, r)


def test_gemm_register_budget():
    """The direct-to-LDS GEMM variants are tuned to a register budget: <= 128 VGPRs (4 workgroups per CU) for the plain /
    fast-conv variants and the fused conv weight gradient, <= 168 (3 per CU) for the generic-conv and 192-column
    variants, no scratch anywhere.  One innocuous edit once pushed the weight-gradient kernel from 122 to 130 VGPRs and
    cost 50 % on 9 ms of the step, silently; the build now records the compiler's allocation and this test reads it."""
    import json
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "a3t_amd", "lib",
                        "gemm_bf16.resources.json")
    if not os.path.exists(path):
        pytest.skip("library not built by a3t_amd.build in this tree")
    rows = json.load(open(path))
    seen = 0
    for name, r in rows.items():
        m = re.match(r"_Z21gemm_bf16_glds_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)EEv2GP", name)
        if not m:
            continue
        layout, stages, wm, conv, wn = map(int, m.groups())
        seen += 1
        assert r["ScratchSize [bytes/lane]"] == 0 and r["VGPRs Spill"] == 0, name
        four_per_cu = wn == 2 and (conv in (0, 1) or (conv == 2 and layout == 2))
        assert r["VGPRs"] <= (128 if four_per_cu else 168), (name, r["VGPRs"])
    assert seen >= 22
    # the persistent 8-phase 256x256 kernel runs one 512-thread workgroup per CU: 256 registers per wave, and NOTHING may
    # spill -- a scratch reload inside its K loop is a vector-memory operation whose wait (vmcnt(0)) drains the DMA pipeline,
    # and spilled SGPRs (the descriptor has ~50 scalar fields) cost VGPR lanes and readlane traffic in every load segment
    path = path.replace("gemm_bf16.resources.json", "gemm_bf16_8p.resources.json")
    rows = json.load(open(path))
    assert len(rows) >= 2
    for name, r in rows.items():
        assert r["ScratchSize [bytes/lane]"] == 0 and r["VGPRs"] <= 256, (name, r["VGPRs"])
        assert r["VGPRs Spill"] == 0 and r["SGPRs Spill"] == 0, (name, r)
    # the 384-column panel kernel: same discipline (one 512-thread workgroup per CU, counted vmcnt waits in the K loop): no
    # scratch.  A few uniform per-tile values may sit in VGPR lanes (v_writelane / v_readlane outside the K loop, no memory):
    # making them opaque inside the issue lambdas, as the 8-phase kernel does, costs this kernel its in-place MFMA accumulators.
    rows = json.load(open(path.replace("gemm_bf16_8p.resources.json", "gemm_bf16_pn.resources.json")))
    assert len(rows) >= 2
    for name, r in rows.items():
        assert r["ScratchSize [bytes/lane]"] == 0 and r["VGPRs"] <= 256, (name, r["VGPRs"])
        assert r["VGPRs Spill"] == 0 and r["SGPRs Spill"] <= 24, (name, r)


def test_sedit_driver_span_arithmetic_matches_reference():
    """a3t_amd.sedit (the product's mirror of espnet2/bin/sedit_inference.py's host logic) against the reference
    driver's own outputs (tests/golden/sedit.json, 40 edits) -- the same checker the oracle is held to -- and against
    the oracle on the data-movement helpers."""
    from test_oracle_golden import check_sedit_impl
    from a3t_amd import sedit
    from oracle import a3t_oracle as O

    def plan(wav, fs, hop, ms, me, op, nph, rep, add, dur, new_str, **opts):
        return sedit.prepare_features_with_duration(wav, fs, hop, ms, me, op, nph, rep, add, dur, new_str, **opts)

    check_sedit_impl(sedit.get_phns_and_spans, plan, sedit.get_masked_mel_boundary, sedit.duration_adjust_factor)
    left, gen, right = torch.randn(1, 3, 4), torch.randn(2, 4), torch.randn(1, 5, 4)
    for parts in ([left, gen, right], [left[:, :0], gen, right], [left, gen, right[:, :0]], [left[:, :0], gen, right[:, :0]]):
        assert torch.equal(sedit.splice_feat_gen(parts), O.sedit_splice_feat_gen(parts))
    a, b = np.arange(3000.0), -np.arange(6000.0)
    assert np.array_equal(sedit.replace_waveform(a, b, 300, [2, 5], [1, 9]), O.sedit_replace_waveform(a, b, 300, [2, 5], [1, 9]))


def test_checkpoint_average_nbest_matches_reference(tmp_path):
    """a3t_amd.checkpoint.average_nbest_models against the reference function's own result on five epoch files
    (tests/golden/average.npz): the 3-best mean bit-exact (same accumulation order), integer entries summed, the same
    files and symlinks; then save_checkpoint / resume round trip with the reference's file names."""
    from a3t_amd import checkpoint as ck
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "average.npz"))
    rep = ck.EpochReport()
    for e, v in g["losses"]:
        e = int(e)
        st = {k.split(".", 2)[2]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"in.{e}.")}
        torch.save(st, tmp_path / f"{e}epoch.pth")
        rep.set_epoch(e)
        rep.register("valid", {"loss": v})
        rep.register("train", {"loss": 2 * v})
    ck.average_nbest_models(tmp_path, rep, [("valid", "loss", "min")], [1, 3, 9])
    assert sorted(p.name for p in tmp_path.iterdir()) == list(g["files"])
    links = sorted(f"{p.name}->{os.readlink(p)}" for p in tmp_path.iterdir() if p.is_symlink())
    assert links == list(g["links"])
    ave = torch.load(tmp_path / "valid.loss.ave_3best.pth")
    for k, v in ave.items():
        ref = g["out.ave3." + k]
        assert v.numpy().dtype == ref.dtype and np.array_equal(v.numpy(), ref), k
    assert int(ave["bn.num_batches_tracked"]) == 200 + 500 + 400          # the three best epochs: 2, 5, 4 (summed, not averaged)

    class M:          # anything with the nn.Module state_dict contract
        def __init__(self):
            self.s = {"w": torch.arange(4.0)}

        def state_dict(self):
            return dict(self.s)

        def load_state_dict(self, s):
            self.s = dict(s)

    m, out = M(), tmp_path / "exp"
    r2 = ck.EpochReport()
    for e, loss in ((1, 0.5), (2, 0.3), (3, 0.4)):
        r2.set_epoch(e)
        r2.register("valid", {"loss": loss})
        m.s["w"] = m.s["w"] + 1
        improved = ck.save_checkpoint(out, m, r2, e)
        assert improved == (["valid.loss"] if e in (1, 2) else [])
    assert os.readlink(out / "latest.pth") == "3epoch.pth" and os.readlink(out / "valid.loss.best.pth") == "2epoch.pth"
    m2, r3 = M(), ck.EpochReport()
    ck.resume(out / "checkpoint.pth", m2, r3)
    assert torch.equal(m2.s["w"], m.s["w"]) and r3.get_epoch() == 3 and r3.stats == r2.stats


def test_masking_plan_is_phones_masking_in_integers():
    """a3t_amd.collate.masking_plan (host half of the on-device collate) consumes the numpy RNG exactly like phones_masking
    and its integer plan, painted the way a3t_collate_paint paints it, gives the same mask -- all four branches of
    collate_fn.py:346-385 (phone spans, span_boundary, mean_phn_span == 0, mlm_prob == 1)."""
    import numpy as np
    from a3t_amd.collate import masking_plan, phones_masking
    rs = np.random.RandomState(3)
    for case in range(40):
        B, T, P = int(rs.randint(1, 5)), int(rs.randint(20, 90)), int(rs.randint(1, 12))
        alen = rs.randint(0, P + 1, size=B)
        fs = np.sort(rs.randint(0, T, size=(B, P)), axis=1).astype(np.int32)
        fe = np.minimum(fs + rs.randint(0, 9, size=(B, P)), T + 3).astype(np.int32)
        flen = rs.randint(T // 2, T + 1, size=B)
        flen[rs.randint(B)] = T
        nonpad = np.arange(T)[None, :] < flen[:, None]
        mode = case % 4
        prob, span, sb = (0.8, 8, None) if mode == 0 else (0.8, 8, np.sort(rs.randint(0, T, size=(B, 4)), axis=1)) if mode == 1 \
            else (0.3, 0, None) if mode == 2 else (1.0, 8, None)
        np.random.seed(case)
        want = phones_masking(T, nonpad, fs, fe, alen, prob, span, sb)
        st_want = np.random.get_state()[1].copy()
        np.random.seed(case)
        sel, mspan, nms = masking_plan(T, alen, prob, span, P, sb)
        assert np.array_equal(np.random.get_state()[1], st_want)          # same draws, same RNG state afterwards
        got = np.zeros((B, T), dtype=bool)
        for b in range(B):
            for j in range(min(int(alen[b]), P)):
                if sel[b, j]:
                    got[b, max(int(fs[b, j]), 0):max(int(fe[b, j]), 0)] = True
            for q in range(int(nms[b])):
                got[b, mspan[b, q, 0]:mspan[b, q, 1]] = True
        got &= nonpad
        assert np.array_equal(got, want), (case, mode)


def test_token_reduction_kernels_keep_their_register_footprint_and_their_dma_pipeline():
    """Two properties of the weight-gradient kernels that only the compiler's output shows (round 5):
    (1) the 128 x 384-tile kernel stays at <= 184 VGPRs: two of its waves per SIMD leave room for a 132-register row kernel of
        the main stream (512 registers per SIMD lane) -- variants at 212-216 registers were 4 % faster alone and 0.4-0.7 ms
        slower inside the step;
    (2) no `s_waitcnt vmcnt(0)` between the first and the last MFMA of either token-reduction kernel: issued through the
        buffer-load-to-LDS builtin, hipcc drained the whole DMA queue in front of every transposed fragment read (7 drains per
        K-loop iteration; FFN weight gradient 173 us instead of 134 with operands in HBM).  The kernels issue their DMA as inline
        asm and wait with counted vmcnt; this test compiles the file to assembly and looks."""
    import json
    import re
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = os.path.join(root, "a3t_amd", "lib", "gemm_bf16_8p.resources.json")
    if not os.path.exists(res):
        pytest.skip("library not built by a3t_amd.build in this tree")
    rows = json.load(open(res))
    tn3 = {k: v for k, v in rows.items() if "tn3_kernel" in k}
    assert len(tn3) == 2
    for name, r in tn3.items():
        assert r["VGPRs"] <= 184 and r["VGPRs Spill"] == 0 and r["SGPRs Spill"] == 0, (name, r["VGPRs"])
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                               "-o", out, os.path.join(root, "a3t_amd", "csrc", "gemm_bf16_8p.hip")], stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\d+gemm_bf16_8p_tn3?_kernelILb[01]E", l)]
    assert len(starts) == 4, [s for _, s in starts]
    for i, name in starts:
        end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith(".Lfunc_end"))
        body = lines[i:end]
        mf = [j for j, l in enumerate(body) if "v_mfma" in l]
        assert len(mf) >= 96, (name, len(mf))
        loop = body[mf[0]:mf[-1]]
        drains = [l.strip() for l in loop if re.search(r"s_waitcnt\s+vmcnt\(0\)", l)]
        counted = [l.strip() for l in loop if re.search(r"s_waitcnt\s+vmcnt\((\d+)\)", l)]
        assert not drains, (name, drains)
        assert len(counted) >= 2, (name, counted)


def test_backward_of_the_legacy_rel_shift_is_a_strided_view_of_the_score_gradient():
    """What the engine's dBD view rests on (a3t_attn_bwd_ds with dbd = NULL, a3t_gemm_desc::a_unaligned), pinned on the CPU against
    autograd through the oracle's restatement of LegacyRelPositionMultiHeadedAttention.rel_shift (attention.py:145-165):
    grad(matrix_bd)[r][c] = grad(scores)_flat[r (T + 1) + c - (T - 1)], i.e. torch.as_strided over the flat score gradient with T zeros
    in front of it (row stride T + 1, one element into the zeros) -- including the T - 1 entries of row 0 that never reach the scores."""
    from oracle import a3t_oracle as O
    g = torch.Generator().manual_seed(0)
    for (B, H, T) in [(1, 1, 8), (2, 3, 40), (1, 2, 137)]:
        bd = torch.randn(B, H, T, T, generator=g, dtype=torch.float64, requires_grad=True)
        ds = torch.randn(B, H, T, T, generator=g, dtype=torch.float64)
        (O.rel_shift_legacy(bd) * ds).sum().backward()
        flat = torch.zeros(B * H, T + T * T, dtype=torch.float64)
        flat[:, T:] = ds.reshape(B * H, T * T)
        view = torch.as_strided(flat, (B * H, T, T), (T + T * T, T + 1, 1), storage_offset=1)
        assert torch.equal(view.reshape(B, H, T, T), bd.grad), (B, H, T)
        assert not bool(bd.grad[:, :, 0, :T - 1].any())
