"""ctypes binding of libmanner_hip.so (the C ABI declared in include/manner_hip.h).

There is no CPU fallback: if the library is missing this raises, and every wrapper in
``manner_amd.hip`` refuses non-GPU tensors.
"""
from __future__ import annotations

import ctypes as C
import os
import re

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "lib", "libmanner_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(PKG), "include", "manner_hip.h")

ABI_VERSION = 8
STATUS_MASK, STATUS_TOKEN, STATUS_FUSED, STATUS_INDEX, STATUS_LENGTHS = 1, 2, 4, 8, 16
PREC_F32, PREC_BF16 = 0, 1
PREC_BF16X3 = 2
PREC_F16 = 3
PREC_F16X3 = 4
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "f16": PREC_F16, "fp16": PREC_F16,
              "f16x3": PREC_F16X3}
W_EMB_COUNT, WL_COUNT = 5, 16
MAX_LEN = 128
PROF_CLASSES = ["lengths", "embed_ln", "gemm_qkv", "attention", "gemm_out", "layernorm", "gemm_ffn1", "gemm_ffn2",
                "gather_cls", "cls_tail"]


class EncoderConfigC(C.Structure):
    _fields_ = [("arch", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32),
                ("intermediate", C.c_int32), ("vocab", C.c_int32), ("max_pos", C.c_int32),
                ("type_vocab", C.c_int32), ("pad_id", C.c_int32), ("ln_eps", C.c_float)]


_P, _I64, _I32, _SZ = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
SIGNATURES = {
    "manner_hip_abi_version": (C.c_int, []),
    "manner_hip_last_error": (C.c_char_p, []),
    "manner_hip_encoder_create": (C.c_int, [C.POINTER(EncoderConfigC), C.POINTER(_P), _I32, C.c_uint32, _P,
                                            C.POINTER(_P)]),
    "manner_hip_encoder_destroy": (C.c_int, [_P]),
    "manner_hip_encoder_workspace_bytes": (_SZ, [_P, _I64, _I64, _I32]),
    "manner_hip_encode_cls": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I32, _P, _P, _SZ, _P]),
    "manner_hip_encode_hidden": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "manner_hip_encoder_status": (C.c_int, [_P, _P]),
    "manner_hip_encoder_status_async": (C.c_int, [_P, _P, _P]),
    "manner_hip_fingerprint": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "manner_hip_encoder_profile": (C.c_int, [_P, _I32]),
    "manner_hip_encoder_profile_read": (C.c_int, [_P, _P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "manner_hip_additive_pool": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I32, _I32, _P, _P, _P]),
    "manner_hip_additive_pool_workspace_bytes": (_SZ, [_I64, _I64, _I32, _I32]),
    "manner_hip_additive_pool_fused": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I32, _I32, _P, _P, _SZ, _I32, _P]),
    "manner_hip_entity_workspace_bytes": (_SZ, [_I64, _I64, _I32]),
    "manner_hip_entity_encode": (C.c_int, [_P, _I64, _I64, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _I32, _P, _P, _SZ, _P, _P]),
    "manner_hip_linear": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "manner_hip_dot": (C.c_int, [_P, _P, _I64, _I64, _I32, _I64, _I64, _I64, _P, _P]),
    "manner_hip_score_late_fusion": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "manner_hip_score_late_fusion_f16": (C.c_int, [_P, _P, _I64, _I32, _P, _P, _P, _P, _I64, _P, _P, _P]),
    "manner_hip_table_to_f16_workspace_bytes": (_SZ, [_I32]),
    "manner_hip_table_to_f16": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _SZ, _P]),
    "manner_hip_score_user": (C.c_int, [_P, _I64, _I32, _P, _P, _P, _I64, _P, _P, _P]),
    "manner_hip_to_dense": (C.c_int, [_P, _P, _I64, _I64, _I32, _P, _P, _P, _P]),
    "manner_hip_news_key128": (C.c_int, [_P, _P, _I64, _I64, _P, _P]),
    "manner_hip_news_cache_lookup": (C.c_int, [_P, _I64, _P, _P, _I64, _P, _I32, _P, _P, _P, _P]),
    "manner_hip_zscore_fuse": (C.c_int, [_P, _I64, _I32, C.POINTER(C.c_float), _P, _I64, _P, _P, _P]),
    "manner_hip_rank_ndcg": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P, _P, _P]),
    "manner_hip_score_fuse_rank_workspace_bytes": (_SZ, [_I32, _I64]),
    "manner_hip_score_fuse_rank": (C.c_int, [C.POINTER(_P), _I32, C.POINTER(C.c_float), _I64, _I32, _P, _P, _P, _P, _I64, _I64, _P, _I32, _P, _P,
                                             _P, _P, _P, _P, _SZ, _P, _P]),
    "manner_hip_aspect_metrics": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _P, _P, _P]),
    "manner_hip_auc_workspace_bytes": (_SZ, [_I64]),
    "manner_hip_auc": (C.c_int, [_P, _P, _I64, _I32, _P, _SZ, _P, _P, _P]),
    "manner_hip_eval_loss": (C.c_int, [_P, _P, _P, _I64, _I32, C.c_float, _I64, _P, _P]),
    "manner_hip_train_saved_bytes": (_SZ, [C.POINTER(EncoderConfigC), _I64, _I64, _I32]),
    "manner_hip_train_saved_bytes_for": (_SZ, [C.POINTER(EncoderConfigC), _I64, _I64, _I32, _I32]),
    "manner_hip_train_workspace_bytes": (_SZ, [C.POINTER(EncoderConfigC), _I64]),
    "manner_hip_train_weight_cache": (C.c_int, [C.POINTER(_P), C.POINTER(_I32), _I32]),
    "manner_hip_train_layout_last": (_I32, []),
    "manner_hip_train_layout_next": (C.c_int, [_I32]),
    "manner_hip_train_forward": (C.c_int, [C.POINTER(EncoderConfigC), C.POINTER(_P), _I32, _P, _P, _I64, _I64, _I64, _I32, _I32, _P,
                                           C.c_float, C.c_float, C.c_float, C.c_uint64, _P, _P, _SZ, _P, _SZ, _P, _P]),
    "manner_hip_train_backward": (C.c_int, [C.POINTER(EncoderConfigC), C.POINTER(_P), _I32, _P, _I64, _I64, _I64, _I32, _I32,
                                            C.c_float, C.c_float, C.c_float, C.c_uint64, _P, _P, _SZ, C.POINTER(_P), _P, _P, _SZ, _P]),
    "manner_hip_train_full_forward": (C.c_int, [C.POINTER(EncoderConfigC), C.POINTER(_P), _I32, _P, _P, _I64, _I64, _I32, C.c_float, C.c_float,
                                                C.c_uint64, _P, _P, _SZ, _P, _SZ, _P, _P]),
    "manner_hip_train_full_backward": (C.c_int, [C.POINTER(EncoderConfigC), C.POINTER(_P), _I32, _P, _I64, _I64, _I32, C.c_float, C.c_float,
                                                 C.c_uint64, _P, _P, _SZ, C.POINTER(_P), _P, _SZ, _P]),
    "manner_hip_dropout_mask": (C.c_int, [C.c_uint64, C.c_uint32, C.c_float, _I64, _P, _P]),
    "manner_hip_late_fusion_train_forward": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "manner_hip_late_fusion_train_backward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "manner_hip_dot_backward": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, _I64, _I64, _I64, _P, _P, _P]),
    "manner_hip_train_loss": (C.c_int, [_P, _P, _P, _I64, _I32, C.c_float, _I64, _P, _P, _P, _P]),
    "manner_hip_supcon_embeddings_workspace_bytes": (_SZ, [_I64]),
    "manner_hip_supcon_embeddings": (C.c_int, [_P, _P, _I64, _I32, C.c_float, _P, _P, _P, _P, _SZ, _P]),
    "manner_hip_linear_backward": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _P]),
    "manner_hip_additive_pool_backward_workspace_bytes": (_SZ, [_I64, _I64, _I32, _I32]),
    "manner_hip_additive_pool_backward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _P, _P, _P, _P, _P, _SZ, _P]),
    "manner_hip_axis0_attention": (C.c_int, [_P, _I64, _I64, _I32, _I32, _P, _P]),
    "manner_hip_axis0_attention_backward": (C.c_int, [_P, _P, _I64, _I64, _I32, _I32, _P, _P, _P]),
    "manner_hip_embedding": (C.c_int, [_P, _I64, _P, _I64, _I32, _P, _P, _P]),
    "manner_hip_embedding_backward": (C.c_int, [_P, _I64, _P, _I64, _I32, _I64, _P, _P]),
    "manner_hip_dropout": (C.c_int, [_P, _P, _I64, C.c_uint64, C.c_uint32, C.c_float, _P]),
    "manner_hip_encode_full_workspace_bytes": (_SZ, [C.POINTER(EncoderConfigC), _I64, _I64]),
    "manner_hip_encode_full": (C.c_int, [C.POINTER(EncoderConfigC), C.POINTER(_P), _I32, _P, _P, _I64, _I64, _I32, _P, _P, _SZ, _P, _P]),
    "manner_hip_mha_axis0_workspace_bytes": (_SZ, [_I64, _I64, _I32]),
    "manner_hip_mha_axis0": (C.c_int, [_P, _I64, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "manner_hip_collate_segments": (C.c_int, [_P, _I64, _I64, _P, _P]),
    "manner_hip_collate_text": (C.c_int, [_P, _P, _I64, _I32, _P, _I64, _I32, _I32, _P, _P, _P]),
    "manner_hip_collate_entities": (C.c_int, [_P, _P, _I64, _I32, _P, _I64, _I32, _P, _P]),
    "manner_hip_collate_aspects": (C.c_int, [_P, _P, _P, _I64, _P, _I64, _P, _P, _P, _P]),
}

_lib = None


def header_symbols():
    """Function names declared in include/manner_hip.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(manner_hip_[a-z_0-9]+)\s*\(", text)))


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MANNeR HIP hot path has no CPU fallback. "
                "Build it with `python -m manner_amd.build` (hipcc, gfx950).")
        # PyTorch-ROCm bundles its own HIP runtime (same soname); let it initialise the device before
        # this library's code objects register with that runtime (the reverse order leaves the
        # library's first hipMalloc failing with hipErrorNoDevice).
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.manner_hip_abi_version() != ABI_VERSION:
            raise RuntimeError("libmanner_hip.so ABI version mismatch; rebuild with `python -m manner_amd.build`")
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"manner_hip error {rc}: {load().manner_hip_last_error().decode(errors='replace')}")
