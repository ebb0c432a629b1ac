"""ctypes binding of liba3t_hip.so (include/a3t_hip.h).  Fails loudly when the HIP library is
missing: there is no CPU fallback anywhere in the product path."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

# torch ships its own libamdhip64.so; it must be the HIP runtime already resident in the process
# when liba3t_hip.so is dlopen'ed, otherwise the kernels would launch on a second, device-less
# runtime (hipErrorNoDevice).  Streams and device pointers come from torch anyway.
import torch  # noqa: F401  (load order matters)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("A3T_LIB_PATH") or os.path.join(HERE, "lib", "liba3t_hip.so")   # (override: instrumented builds of tools/)

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SWISH = 0, 1, 2, 3
ACC_STORE, ACC_ADD, ACC_ATOMIC, ACC_SOLE = 0, 1, 2, 3


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("R", c_void_p), ("S", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("a_rs", c_int64), ("a_cs", c_int64),
        ("b_rs", c_int64), ("b_cs", c_int64), ("b_ts", c_int64),
        ("c_rs", c_int64),
        ("batch", c_int32), ("batch_inner", c_int32),
        ("a_bs0", c_int64), ("a_bs1", c_int64), ("b_bs0", c_int64), ("b_bs1", c_int64),
        ("c_bs0", c_int64), ("c_bs1", c_int64),
        ("taps", c_int32), ("pad", c_int32), ("dil", c_int32), ("Tseq", c_int32), ("kshift", c_int32),
        ("alpha", c_float),
        ("act", c_int32), ("accumulate", c_int32), ("splitk", c_int32),
        ("a_dtype", c_int32), ("b_dtype", c_int32), ("c_dtype", c_int32), ("compute", c_int32),
        ("s_dtype", c_int32), ("colsum_slots", c_int32),
        ("colsum", c_void_p), ("colsum_bs1", c_int64), ("colsum_scale", c_float), ("drop_key", ctypes.c_uint32),
        ("drop_p", c_float), ("colsum_ss", c_int32),
        ("keep_out", c_void_p), ("keep_in", c_void_p),
        ("a_signmask", c_int32), ("keep_layout", c_int32),
        ("A2", c_void_p), ("B2", c_void_p), ("b2_cs", c_int64), ("b2_bs0", c_int64), ("b2_bs1", c_int64), ("colsum2", c_void_p),
        ("a2_rs", c_int64), ("a_unaligned", c_int32),
    ]


_P = c_void_p
_SIGS = {
    "a3t_gemm": [POINTER(GemmDesc), _P],
    "a3t_layernorm_fwd": [_P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_float, _P],
    "a3t_layernorm_bwd": [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_int, c_int, c_float,
                          ctypes.c_uint32, _P],
    "a3t_col_reduce": [_P, c_int, _P, _P, _P, _P, c_int, c_int, c_int64, c_int, _P],
    "a3t_f64_to_f32_add": [_P, _P, c_int, c_float, _P],
    "a3t_bn_act_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, c_int, _P],
    "a3t_bn_act_bwd_a": [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "a3t_bn_act_bwd_b": [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "a3t_glu_dwconv_fwd": [_P, c_int, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P],
    "a3t_glu_dwconv_bwd": [_P, _P, c_int, _P, c_int, _P, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "a3t_add_pos_bias": [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "a3t_add_pos_bias_bwd": [_P, _P, _P, c_int, c_int, c_int, _P],
    "a3t_relpos_softmax_fwd": [_P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_float, _P,
                               c_float, ctypes.c_uint32, _P],
    "a3t_relpos_softmax_bwd": [_P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_float, _P,
                               c_float, c_int64, c_int64, ctypes.c_uint32, _P, _P],
    "a3t_attn_fwd_train": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64,
                           c_int64, c_float, c_float, ctypes.c_uint32, _P, _P, _P],
    "a3t_attn_scale_rows": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "a3t_attn_bwd_ds": [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_float,
                        c_float, ctypes.c_uint32, c_int, c_int64, _P],
    "a3t_attn_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
                     c_float, c_float, ctypes.c_uint32, _P, _P, _P],
    "a3t_pwg_block": [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "a3t_mask_fill": [_P, _P, _P, _P, c_int, c_int, c_int, _P],
    "a3t_embed_finish_fwd": [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, ctypes.c_uint32,
                             _P, _P],
    "a3t_embed_finish_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                             ctypes.c_uint32, _P],
    "a3t_scale": [_P, _P, c_int64, c_float, _P],
    "a3t_axpy": [_P, _P, c_int64, c_float, _P],
    "a3t_attn_bias_fold": [_P, c_int32, c_int32, _P, _P, _P, _P],
    "a3t_scale_dev": [_P, _P, c_int64, _P, _P],
    "a3t_slice_rows": [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "a3t_cast_bf16": [_P, _P, c_int64, _P],
    "a3t_cast_bf16_conv_t": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "a3t_split_bf16": [_P, _P, _P, c_int64, _P],
    "a3t_reflect_pad": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "a3t_stft_amp": [_P, _P, c_int64, c_int, c_int, _P],
    "a3t_logmel_finish": [_P, _P, c_int, c_int, c_int, _P],
    "a3t_mlm_loss": [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P],
    "a3t_mlm_loss_scratch_floats": [c_int],
    "a3t_sumsq": [_P, c_int64, _P, _P],
    "a3t_clip_adam": [_P, _P, _P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int, c_float, c_float,
                      _P],
    "a3t_clip_adam_noam": [_P, _P, _P, _P, _P, _P, c_int64, _P, c_float, c_float, c_float, c_float, c_float, c_float,
                           c_float, c_float, _P],
    "a3t_pwg_gate": [_P, _P, _P, c_int64, c_int, _P],
    "a3t_pwg_res_skip": [_P, _P, _P, c_int64, c_int, c_int, _P],
    "a3t_pwg_upsample": [_P, _P, _P, c_int64, c_int64, c_int, c_int, _P],
    "a3t_replicate_pad": [_P, _P, c_int64, c_int64, c_int, c_int, _P],
    "a3t_bias_act": [_P, _P, c_int64, c_int, c_int, c_float, _P],
    "a3t_dropout": [_P, c_int, _P, c_int, c_int64, c_float, ctypes.c_uint32, c_float, _P],
    "a3t_collate_paint": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "a3t_segment_colsum": [_P, _P, c_int, c_int, c_int, _P],
    "a3t_gemm_8p_mode": [c_int],
    "a3t_gemm_8p_supported": [c_int, c_int, c_int, c_int, c_int],
    "a3t_gemm_pn_mode": [c_int],
    "a3t_gemm_tt_mode": [c_int],
    "a3t_gemm_tn3_mode": [c_int],
    "a3t_gemm_tn3_group": [_P, c_int, _P],
    "a3t_attn_split_mode": [c_int],
    "a3t_release_workspaces": [],
    "a3t_gemm_pn_supported": [c_int, c_int, c_int, c_int, c_int],
    "a3t_gemm_tt_supported": [c_int, c_int, c_int, c_int],
    "a3t_dropout_bwd_cast": [_P, _P, c_int, _P, c_float, c_int, c_int, c_float, ctypes.c_uint32, _P],
}
EXPORTS = sorted(list(_SIGS) + ["a3t_version", "a3t_gemm_last_kernel", "a3t_gemm_keep_bytes"])

_lib = None


class A3TLibraryError(RuntimeError):
    pass


def load():
    """Load liba3t_hip.so (built by a3t_amd/build.py).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise A3TLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m a3t_amd.build` (hipcc --offload-arch=gfx950). "
            "a3t_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int
    lib.a3t_version.restype = c_char_p
    lib.a3t_version.argtypes = []
    lib.a3t_gemm_last_kernel.restype = c_char_p
    lib.a3t_gemm_last_kernel.argtypes = []
    lib.a3t_gemm_keep_bytes.restype = c_int64
    lib.a3t_gemm_keep_bytes.argtypes = [c_int, c_int]
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise A3TLibraryError(f"liba3t_hip call failed ({what}): rc={rc}")
