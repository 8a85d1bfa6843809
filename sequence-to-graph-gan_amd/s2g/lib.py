"""ctypes binding of libs2g.so (the C ABI declared in include/s2g.h).

The library is the product path: if it is missing or a symbol is absent the import fails
loudly -- there is no PyTorch fallback for any of these entry points."""
from __future__ import annotations

import ctypes
import os

# torch ships its own HIP runtime (libamdhip64); it must be the first one loaded so that
# libs2g.so binds to the same runtime instance that owns torch's streams and allocations.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# S2G_LIB points kernel-tuning tools at an alternative build of the same library
LIB_PATH = os.environ.get("S2G_LIB") or os.path.join(_HERE, "libs2g.so")

P, I, L, F, U64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint64


class S2GSeq(ctypes.Structure):
    _fields_ = [("len", I), ("tok_off", I), ("p_off", I), ("p_len", I), ("ca", F), ("cp", F),
                ("mask_off", I), ("ws_row", I)]


class S2GGemm(ctypes.Structure):
    _fields_ = [("A", P), ("sam", L), ("sak", L), ("B", P), ("sbk", L), ("sbn", L), ("C", P), ("ldc", L),
                ("slab", L), ("bias", P), ("M", I), ("N", I), ("K", I), ("kchunk", I), ("segT", I),
                ("a_seg", I), ("b_seg", I), ("shift", I), ("accumulate", I), ("act", I), ("alpha", F), ("bscale", P)]


class S2GHead(ctypes.Structure):
    _fields_ = [("hT", P), ("W", P), ("bias", P), ("labels", P), ("B", I), ("mode", I), ("inv_n", F),
                ("inv_wsum", F), ("w0", F), ("w1", F)]


class S2GAttn(ctypes.Structure):
    _fields_ = [("N", I), ("H", I), ("T", I), ("D", I), ("ld", L), ("ldo", L), ("scale", F)]


class S2GLstm64Stack(ctypes.Structure):
    _fields_ = ([("pin0", P)] + [(k, P * 3) for k in ("whh", "wih", "bsum", "gates", "cst", "hout", "mask", "hdrop")]
                + [("tokens", P), ("tab", P), ("labels", P), ("tabl", P), ("V", I), ("n_classes", I)])


class S2GLstm64StackBwd(ctypes.Structure):
    _fields_ = [("dh_top", P), ("whh", P * 2), ("wih1", P), ("gates", P * 2), ("cst", P * 2), ("mask0", P),
                ("delta", P * 2), ("bias_part", P * 2)]


class S2GReduce(ctypes.Structure):
    _fields_ = [("slabs", P), ("slab", L), ("out", P), ("nslab", I), ("rows", I), ("cols", I)]


class S2GCriticUpdate(ctypes.Structure):
    _fields_ = [("job", S2GReduce * 4), ("poff", L * 4), ("p", P), ("g", P), ("m", P), ("v", P),
                ("o_bih0", L), ("o_bhh0", L), ("o_bih1", L), ("o_bhh1", L), ("o_fcw", L), ("o_fcb", L),
                ("n_fcw", I), ("n_fcb", I), ("V", I), ("Vp", I), ("WT0", P), ("W0p", P), ("b0", P), ("b1", P),
                ("lr", F), ("b1c", F), ("b2c", F), ("eps", F), ("step", I)]


# name -> argument types (every function returns int and takes the stream last)
SIGNATURES = {
    "s2g_version": [],
    "s2g_gumbel_softmax_fwd": [P, P, P, I, P, P, L, I, F, I, I, P],
    "s2g_gumbel_softmax_fwd_rng": [P, P, I, P, P, L, I, F, I, U64, U64, P],
    "s2g_rng_fill": [P, L, I, U64, U64, P],
    "s2g_dropout_mask_rng": [P, L, I, I, F, U64, U64, P],
    "s2g_gumbel_softmax_bwd": [P, P, P, L, I, F, P],
    "s2g_argmax_rows": [P, P, L, I, P],
    "s2g_gemm_f32": [ctypes.POINTER(S2GGemm), I, P],
    "s2g_gemm_tall": [P, L, P, L, P, L, I, I, I, P],
    "s2g_gemm_tall_bf16": [P, L, P, L, P, L, I, I, I, P],
    "s2g_gemm_grouped": [P, P, I, I, I, P],
    "s2g_gemm_grouped_bf16": [P, P, I, I, I, P],
    "s2g_reduce_slabs_weighted": [P, L, I, P, P, L, P],
    "s2g_generate_tokens": [P, P, P, P, P, I, I, I, I, U64, U64, P],
    "s2g_linear_small_batch": [P, P, P, P, I, I, I, I, P],
    "s2g_linear_wgrad_adam": [P, P, P, P, P, P, I, I, I, F, F, F, F, I, I, P],
    "s2g_small_linear_grads": [I, P, P, P, P, P, P, I, P],
    "s2g_critic_update": [ctypes.POINTER(S2GCriticUpdate), P],
    "s2g_reduce_slabs": [P, L, I, I, I, P, I, I, F, P],
    "s2g_reduce_slabs_multi": [P, I, I, I, F, P],
    "s2g_colsum": [P, L, L, I, P, L, P],
    "s2g_bias_sums": [P, P, L, P, P, I, P, P, P, P, P, I, P],
    "s2g_bias_sums_noalloc": [P, P, L, P, I, P, P, P, P, P, I, P],
    "s2g_bias_sums_final": [P, I, L, P, I, P, P, P, P, P, I, P],
    "s2g_lstm2_fwd": [P, I, I, I] + [P] * 15,
    "s2g_lstm2_bwd": [P, I, I] + [P] * 13 + [I, I, P, P, P],
    "s2g_lstm2_tangent": [P, I] + [P] * 14 + [P, F, F, P],
    "s2g_lstm2_bwd2": [P, I, I] + [P] * 18,
    "s2g_pack_critic": [P] * 9 + [I, P],
    "s2g_critic_batch_desc": [P, P, I, I, I, I, P, P],
    "s2g_head_fwd": [P, P, P, P, I, I, P],
    "s2g_head_bwd_dh": [P, P, P, I, I, P],
    "s2g_head_bwd_w": [P, P, P, P, I, I, I, P],
    "s2g_critic_loss": [P, P, I, F, F, F, F, P, P, P],
    "s2g_generator_loss": [P, P, I, F, F, F, F, P, P, P],
    "s2g_critic_head": [P, P, P, P, I, F, F, F, F, P, P, P, P, P, P, P],
    "s2g_one_hot": [P, P, I, P, L, I, P],
    "s2g_gp_interp": [P, P, P, P, I, I, I, I, P, P],
    "s2g_gp_interp_dense": [P, P, P, P, I, I, I, I, P, P],
    "s2g_gp_norm_penalty": [P, I, L, F, F, P, P, I, P],
    "s2g_gp_norm_penalty_proj": [P, P, I, L, F, F, P, P, P],
    "s2g_gp_dots": [P, P, I, L, P, P],
    "s2g_gp_penalty_final": [P, I, F, F, P, P, P, P],
    "s2g_gp_scale": [P, P, P, F, L, P, I, P],
    "s2g_embed_mul": [P, P, P, P, I, I, I, P],
    "s2g_embed_mul_bwd": [P, P, P, P, I, I, I, P],
    "s2g_lrelu_bwd": [P, P, L, P],
    "s2g_relu_bwd": [P, P, L, P],
    "s2g_add_layernorm_fwd": [P, P, P, P, P, F, P, P, P, L, I, P],
    "s2g_add_layernorm_bwd": [P, P, P, P, P, P, L, I, P],
    "s2g_dropout_mask": [P, L, F, P],
    "s2g_bn_stats": [P, P, I, I, P],
    "s2g_bn_lrelu_apply": [P, P, F, P, P, P, P, F, F, I, P, P, P, I, I, P],
    "s2g_bn_lrelu_bwd_stats": [P, P, P, P, I, I, P],
    "s2g_bn_bwd_apply": [P, P, P, F, P, P, P, I, I, P],
    "s2g_bn_lrelu_fwd_fused": [P, P, P, P, P, F, F, I, P, P, P, I, I, P],
    "s2g_bn_lrelu_bwd_fused": [P, P, P, P, P, P, P, P, I, I, P],
    "s2g_adam": [P, P, P, P, L, F, F, F, F, I, F, P],
    "s2g_lstm64_fwd": [P, P, P, I, I, P, P, P, P, P, P],
    "s2g_lstm64_bwd": [P, P, P, I, I, P, P, P, P],
    "s2g_lstm64_stack_fwd": [ctypes.POINTER(S2GLstm64Stack), I, P, I, I, P],
    "s2g_lstm64_stack_bwd": [ctypes.POINTER(S2GLstm64StackBwd), P, I, I, P],
    "s2g_embed_concat": [P, P, I, P, P, I, I, L, I, P, P],
    "s2g_embed_grad_tokens_scratch": [L, I],
    "s2g_embed_grad_tokens": [P, P, I, L, I, P, P, I, P, P],
    "s2g_embed_grad_labels": [P, P, I, I, I, P, P, I, P, P],
    "s2g_bucket_rows_by_token": [P, P, I, L, I, P, I, I, P, P, P],
    "s2g_bucket_rows_by_token_t": [P, P, I, L, I, P, I, I, I, P, I, P, P],
    "s2g_bucket_rows_by_label": [P, I, I, P, I, P, P],
    "s2g_attnpool_fwd": [P, P, P, P, P, I, I, I, P, P, P, P],
    "s2g_attnpool_bwd": [P, P, P, P, P, P, I, I, I, P, P, P, P, P],
    "s2g_wasserstein_loss": [P, I, P, P, P],
    "s2g_clamp": [P, L, F, P],
    "s2g_mul_inplace": [P, P, L, P],
    "s2g_add_inplace": [P, P, L, P],
    "s2g_attention_fwd": [P, P, P, P, P, P, P, ctypes.POINTER(S2GAttn), P],
    "s2g_attention_bwd": [P, P, P, P, P, P, P, P, P, P, P, ctypes.POINTER(S2GAttn), P],
    "s2g_graph_features": [P, P, I, I, I, P, P, P, L, P],
    "s2g_graph_features_workspace": [I, I],
    "s2g_stream_create_cu_mask": [P, I, P],
    "s2g_stream_destroy": [P],
}


class S2GError(RuntimeError):
    pass


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise S2GError(f"{path} not found: build it with `make -C sequence-to-graph-gan_amd/csrc` "
                       "(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise S2GError(f"libs2g.so does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.s2g_graph_features_workspace.restype = ctypes.c_long      # a byte count
    lib.s2g_embed_grad_tokens_scratch.restype = ctypes.c_long     # a float count
    return lib


_LIB = None


def get() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = load()
    return _LIB


def call(name: str, *args) -> None:
    rc = getattr(get(), name)(*args)
    if rc != 0:
        raise S2GError(f"{name} failed with code {rc}")
