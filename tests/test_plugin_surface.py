"""CPU tests of the drop-in boundary: Task / Model plugin surface and checkpoint format."""
import argparse
import copy
import os

import numpy as np
import pytest
import torch
import yaml

from a3t_amd.task import MLMTask

RECIPE_ENC = dict(input_layer="sega_mlm", pre_speech_layer=0, cnn_module_kernel=7, attention_dim=32, attention_heads=2,
                  linear_units=64, num_blocks=1, dropout_rate=0.2, positional_dropout_rate=0.2,
                  attention_dropout_rate=0.2, normalize_before=True, macaron_style=True, use_cnn_module=True,
                  selfattention_layer_type="rel_selfattn", activation_type="swish", pos_enc_layer_type="rel_pos",
                  positionwise_layer_type="conv1d", positionwise_conv_kernel_size=3)
RECIPE_DEC = dict(cnn_module_kernel=31, attention_dim=32, attention_heads=2, linear_units=64, num_blocks=1,
                  dropout_rate=0.2, positional_dropout_rate=0.2, attention_dropout_rate=0.2, macaron_style=True,
                  use_cnn_module=True, selfattention_layer_type="rel_selfattn", activation_type="swish",
                  pos_enc_layer_type="rel_pos", positionwise_layer_type="conv1d", positionwise_conv_kernel_size=3)
MODEL_CONF = dict(lsm_weight=0.1, length_normalized_loss=False, masking_schema="phn_span", mean_phn_span=8,
                  mlm_prob=0.8, dynamic_mlm_prob=False, postnet_layers=2, postnet_filts=5, postnet_chans=16)


def _args():
    return argparse.Namespace(token_list=["<blank>", "<unk>"] + [f"p{i}" for i in range(8)] + ["<sos/eos>"], odim=80,
                              input_size=80, feats_extract="fbank",
                              feats_extract_conf=dict(n_fft=2048, hop_length=300, win_length=1200, fs=24000, fmin=80,
                                                      fmax=7600, n_mels=80),
                              normalize=None, normalize_conf={}, use_scaled_pos_enc=False, encoder="conformer",
                              encoder_conf=copy.deepcopy(RECIPE_ENC), decoder="conformer",
                              decoder_conf=copy.deepcopy(RECIPE_DEC), model_conf=copy.deepcopy(MODEL_CONF),
                              init="xavier_uniform")


def test_build_model_mutates_args_like_reference_and_exposes_reference_state_dict():
    from oracle import a3t_oracle as O
    args = _args()
    model = MLMTask.build_model(args)
    assert args.encoder_conf["pos_enc_layer_type"] == "legacy_rel_pos"
    assert args.decoder_conf["selfattention_layer_type"] == "legacy_rel_selfattn"
    assert args.feats_extract is None and args.feats_extract_conf is None      # odim given -> features from loader
    ref_shapes = O.param_shapes(O.tiny_config())
    sd = model.state_dict()
    assert set(sd.keys()) == set(ref_shapes.keys())
    for k, shp in ref_shapes.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert model.encoder._output_size == 32 and model.odim == 80 and model.mlm_prob == 0.8
    assert MLMTask.required_data_names() == ("speech",)
    assert MLMTask.optional_data_names() == ("text", "align_start", "align_end")
    assert sum(p.numel() for p in model.parameters()) == model.store.n_params


def test_checkpoint_roundtrip_and_embed_rename(tmp_path):
    from oracle import a3t_oracle as O
    args = _args()
    model = MLMTask.build_model(args)
    state = O.procedural_state(O.param_shapes(O.tiny_config()), 5)
    ck = {k.replace("encoder.speech_embed", "encoder.embed"): torch.from_numpy(np.array(v)) for k, v in state.items()}
    torch.save(ck, tmp_path / "1epoch.pth")
    conf = {k: v for k, v in vars(_args()).items()}
    conf["model_conf"]["ctc_weight"] = 0.0
    with open(tmp_path / "config.yaml", "w") as f:
        yaml.safe_dump(conf, f)
    m2, a2 = MLMTask.build_model_from_file(None, tmp_path / "1epoch.pth", "cpu")
    sd = m2.state_dict()
    for k, v in state.items():
        np.testing.assert_array_equal(sd[k].numpy(), v, err_msg=k)


def test_forward_without_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from oracle import a3t_oracle as O
    model = MLMTask.build_model(_args())
    b = O.synthetic_batch(O.tiny_config(), 2, 48, 8, seed=1)
    with pytest.raises(Exception):
        model(**b)
