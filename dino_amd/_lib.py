"""ctypes binding of libdino_hip.so (the C ABI declared in include/dino_hip.h).

The product path fails loudly when the library is missing or a call returns an error:
there is no CPU / PyTorch fallback.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdino_hip.so")

DINO_F32, DINO_BF16 = 0, 1
DINO_CHUNK = 8192
ABI_VERSION = 10


class GemmArgs(ctypes.Structure):
    _fields_ = [
        ("dtype", c_int32), ("trans", c_int32),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("splitk", c_int32),
        ("A", c_void_p), ("lda", c_int64),
        ("B", c_void_p), ("ldb", c_int64),
        ("out", c_void_p), ("ldo", c_int64),
        ("out_mode", c_int32),
        ("act", c_int32),
        ("bias", c_void_p),
        ("aux", c_void_p), ("ldaux", c_int64),
        ("residual", c_void_p), ("ldres", c_int64),
        ("rowscale", c_void_p), ("rows_per_group", c_int32),
        ("bias_rowscale", c_int32), ("out_rowscale", c_int32),
        ("row_map_n", c_int32),
        ("res_mod", c_int32),
        ("colsum", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_int64),
        ("prescale", c_void_p), ("ln_stats", c_void_p),
        ("ln_stats_in", c_void_p), ("ln_mean_out", c_void_p), ("ln_rstd_out", c_void_p),
        ("ln_dim", c_int32), ("ln_eps", c_float),
    ]


P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> argtypes; every function returns int (0 = OK)
SIGNATURES = {
    "dino_abi_version": [],
    "dino_diag_occupy_cus": [I, I, P],
    "dino_gemm_work_queue": [I],
    "dino_gemm": [ctypes.POINTER(GemmArgs), P],
    "dino_gemm_group": [ctypes.POINTER(GemmArgs), I, P],
    "dino_sum_slabs": [P, I, L, P, I, P],
    "dino_layernorm_fwd": [I, P, L, P, L, P, P, P, P, L, I, F, P],
    "dino_layernorm_bwd": [I, P, L, P, L, P, P, P, P, L, P, L, P, P, P, P, I, L, I, P, L, P],
    "dino_layernorm_bwd_xn": [I, P, L, P, L, P, P, P, P, P, L, P, L, P, L, P, P, P, P, I, L, I, P, L, P],
    "dino_gemm_wide_tile_supported": [ctypes.POINTER(GemmArgs)],
    "dino_ln_stats_finalize": [P, I, L, I, F, P, P, P],
    "dino_layernorm_stats": [I, P, L, P, P, L, I, F, P],
    "dino_ln_fold_weights": [P, P, I, P, P, I, P],
    "dino_layernorm_bwd_static": [I, P, L, P, L, P, P, P, P, L, P, L, P, P, P, P, I, L, I, P],
    "dino_attn_fwd": [I, P, P, P, I, I, I, P, P],
    "dino_attn_probs": [I, P, P, I, I, I, P],
    "dino_attn_cls_fwd": [I, P, P, P, I, I, I, P, P],
    "dino_attn_cls_bwd": [I, P, P, P, P, I, I, I, P],
    "dino_attn_bwd": [I, P, P, P, P, P, I, I, I, P, P, P],
    "dino_loss_fwd_bwd": [I, P, P, P, P, P, P, P, I, I, I, F, F, P],
    "dino_center_update": [P, P, I, F, F, P, I, P, P],
    "dino_proto_loss_chunk": [I, I, P, L, P, L, P, P, P, I, I, I, I, I, F, F, P, P, P],
    "dino_grad_sqnorm": [P, P, I, P, P],
    "dino_clip_coef": [P, P, I, P, F, P, P, P, P],
    "dino_grad_scale": [P, P, I, P, P],
    "dino_adamw_ema": [P, P, P, P, P, P, P, P, I, P, P, F, F, F, F, F, P],
    "dino_ema": [P, P, P, L, F, F, P],
    "dino_cast_bf16": [P, P, L, P],
    "dino_transpose_batched": [I, P, P, P, I, P],
    "dino_im2col": [I, P, P, I, I, I, I, I, P],
    "dino_to_tensor_normalize": [P, P, L, I, I, I, I, P, P, P],
    "dino_drop_path_scales": [ctypes.c_uint64, P, I, I, P, L, P, P, P],
    "dino_aug_resized_crop": [P, I, I, I, P, I, I, I, P, P, P, I, P, P],
    "dino_aug_gray_sums": [P, I, I, P, P, I, P],
    "dino_aug_color_op": [P, I, I, P, I, P, P],
    "dino_aug_box_blur": [P, P, I, I, P, I, P],
    "dino_pos_table": [I, P, P, P, I, P, I, I, P],
    "dino_cls_rows": [I, P, P, P, I, I, I, P],
    "dino_fold_rows": [I, P, P, I, I, I, P],
    "dino_pos_table_bwd": [P, P, P, I, P, I, I, P],
    "dino_l2norm_fwd": [I, P, P, P, L, I, P],
    "dino_l2norm_bwd": [I, P, P, P, P, L, I, P],
    "dino_weightnorm_fwd": [I, P, P, P, P, P, I, I, P],
    "dino_weightnorm_bwd": [P, P, P, P, P, P, I, I, P],
    "dino_colsum": [I, P, L, P, L, I, P, I, P],
    "dino_scale_by_scalar": [I, P, L, P, P],
}

_lib = None


class DinoHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DinoHipError(
            f"{LIB_PATH} not found: build it with `make -C dino_amd/csrc` "
            "(or __graft_entry__.build()); dino_amd has no non-HIP fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = c_int
    v = lib.dino_abi_version()
    if v != ABI_VERSION:
        raise DinoHipError(f"libdino_hip.so ABI {v} != binding ABI {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code == 0:
        return
    if code < 0:
        msg = {-1: "bad argument", -2: "unsupported shape",
               -3: "size beyond the capacity of the allocation-free entry point"}.get(code, "error")
        raise DinoHipError(f"{what}: {msg} (code {code})")
    raise DinoHipError(f"{what}: HIP error {code}")
