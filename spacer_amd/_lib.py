"""ctypes binding of libspacer_hip.so (the C-ABI declared in include/spacer_hip.h).

The product path has NO CPU fallback: if the shared library is missing the import of anything that needs
it raises, loudly.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C spacer_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspacer_hip.so")

SPACER_ACT_NONE, SPACER_ACT_QUICK_GELU, SPACER_ACT_GELU_ERF, SPACER_ACT_SILU = 0, 1, 2, 3
SPACER_PAIR_SWIGLU, SPACER_PAIR_ROPE, SPACER_PAIR_ACT = 2, 3, 4          # enum spacer_pair_epilogue


class SpacerError(RuntimeError):
    pass


class Plan(C.Structure):
    """include/spacer_hip.h: spacer_plan -- launch-plan switches handed to the library explicitly (it reads no environment)."""
    _fields_ = [("struct_bytes", C.c_int), ("gemm_tile", C.c_int), ("gemm_no_split", C.c_int), ("skinny_blocks", C.c_int),
                ("skinny_no_balance", C.c_int), ("cus", C.c_int), ("skinny_skew", C.c_int)]

    def __init__(self, **switches):
        super().__init__(**switches)
        self.struct_bytes = C.sizeof(Plan)      # the library rejects a plan of another size (a stale, shorter binding)


class GemmEpilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_long), ("out_f32", C.c_int),
                ("act", C.c_int), ("alpha", C.c_float), ("workspace", C.c_void_p), ("workspace_bytes", C.c_long),
                ("plan", C.POINTER(Plan))]


class AttnSegment(C.Structure):
    _fields_ = [("q_start", C.c_int), ("q_len", C.c_int), ("pre_start", C.c_int), ("pre_len", C.c_int)]


_p, _i, _l, _f, _u64 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64

# name -> argtypes; every entry returns int.  Kept in the order of include/spacer_hip.h.
SIGNATURES = {
    "spacer_gemm_bf16_nt": [_p, _l, _p, _l, _p, _l, _i, _i, _i, C.POINTER(GemmEpilogue), _p],
    "spacer_gemm_bf16": [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _i, C.POINTER(GemmEpilogue), _p],
    "spacer_gemm_bf16_pair_nt": [_p, _p, _l, _p, _l, _p, _l, _i, _i, _i, C.POINTER(GemmEpilogue), _p],
    "spacer_gemm_bf16_pair_epilogue": [_i, _p, _p, _l, _p, _l, _p, _p, _p, _l, _p, _l, _p, _p, _i, _i, _i, _i, _i, _i, _p, _l, C.POINTER(Plan), _p],
    "spacer_gemm_skinny_bf16": [_p, _l, _p, _l, _p, _l, _i, _i, _i, C.POINTER(GemmEpilogue), _p],
    "spacer_pack_weight_frag": [_p, _l, _p, _i, _i, _p],
    "spacer_gemm_skinny_packed_bf16": [_p, _l, _p, _p, _l, _i, _i, _i, C.POINTER(Plan), _p],
    "spacer_gemm_skinny_packed_store_bf16": [_p, _l, _p, _p, _l, _i, _i, _i, C.POINTER(Plan), _p],
    "spacer_pack_weight_frag_swiglu": [_p, _l, _p, _i, _i, _p],
    "spacer_gemm_skinny_swiglu_bf16": [_p, _l, _p, _p, _l, _i, _i, _i, _p],
    "spacer_gemm_skinny_swiglu_bf16_ws": [_p, _l, _p, _p, _l, _i, _i, _i, _p, _l, C.POINTER(Plan), _p],
    "spacer_transpose_bf16": [_p, _l, _p, _l, _i, _i, _i, _p],
    "spacer_rmsnorm_fwd": [_p, _i, _p, _p, _p, _i, _i, _f, _p],
    "spacer_rmsnorm_bwd": [_p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _p],
    "spacer_rmsnorm_bwd_ws": [_p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _p, _l, _p],
    "spacer_layernorm_fwd": [_p, _i, _p, _p, _p, _p, _p, _i, _i, _f, _p],
    "spacer_layernorm_bwd": [_p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _p],
    "spacer_layernorm_bwd_ws": [_p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _p, _l, _p],
    "spacer_rope_inplace": [_p, _l, _p, _p, _i, _i, _i, _i, _p],
    "spacer_attn_fwd": [_p, _p, _p, _p, _p, _l, _l, _l, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "spacer_attn_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _l, _l, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "spacer_attn_decode": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p],
    "spacer_attn_decode_shared": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "spacer_attn_decode_shared_rows": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "spacer_swiglu_fwd": [_p, _p, _i, _i, _p],
    "spacer_swiglu_bwd": [_p, _p, _p, _i, _i, _p],
    "spacer_act_fwd": [_p, _p, _l, _i, _p],
    "spacer_act_bwd": [_p, _p, _p, _l, _i, _p],
    "spacer_bias_grad": [_p, _l, _p, _i, _i, _p],
    "spacer_zero": [_p, _l, _p],
    "spacer_cast_f32_to_bf16": [_p, _p, _l, _p],
    "spacer_cast_bf16_to_f32": [_p, _p, _l, _p],
    "spacer_cast_f32_to_bf16_strided": [_p, _l, _p, _l, _i, _i, _p],
    "spacer_gather_rows_bf16": [_p, _l, _p, _p, _i, _i, _p],
    "spacer_scatter_add_rows_f32": [_p, _p, _p, _l, _i, _i, _p],
    "spacer_embed_fwd": [_p, _p, _p, _p, _p, _i, _i, _p],
    "spacer_embed_bwd": [_p, _p, _p, _p, _p, _i, _i, _p],
    "spacer_patchify": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "spacer_logprob_fwd": [_p, _l, _p, _p, _p, _i, _i, _p],
    "spacer_logprob_bwd": [_p, _l, _p, _p, _p, _p, _l, _i, _i, _p],
    "spacer_lse_chunk": [_p, _l, _p, _i, _i, _p, _p, _p, _i, _i, _p],
    "spacer_lse_finish": [_p, _p, _p, _p, _p, _i, _p],
    "spacer_logprob_bwd_chunk": [_p, _l, _p, _i, _p, _p, _p, _l, _i, _i, _p],
    "spacer_grpo_loss": [_p, _p, _p, _p, _f, _p, _p, _p, _i, _i, _p],
    "spacer_completion_mask": [_p, _i, _p, _p, _i, _i, _p],
    "spacer_gather_f32": [_p, _p, _p, _l, _p],
    "spacer_scatter_f32": [_p, _p, _p, _l, _p],
    "spacer_sample_top_p": [_p, _l, _i, _i, _i, _f, _f, _u64, _p, _i, _i, _i, _p, _p, _p, _p, _l, _p],
    "spacer_sample_top_p_step": [_p, _l, _i, _i, _i, _f, _f, _u64, _p, _i, _i, _i, _i, _p, _p, _p, _l, _p],
    "spacer_sample_top_p_step_ws": [_p, _l, _i, _i, _i, _f, _f, _u64, _p, _i, _i, _i, _i, _p, _p, _p, _l, _p, _l, _p],
    "spacer_decode_embed": [_p, _p, _p, _i, _i, _p, _p, _p],
    "spacer_eos_schedule": [_p, _l, _i, _i, _p, _i, _p, _i, _p],
    "spacer_decode_rope_table": [_p, _p, _f, _p, _p, _i, _i, _p],
    "spacer_decode_qkv_finish": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "spacer_decode_qkv_finish_normed": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _i, _i, _i, _i, _i, _p],
    "spacer_gemm_skinny_packed_normed": [_p, _l, _p, _p, _l, _p, _i, _i, _i, C.POINTER(Plan), _p],
    "spacer_gemm_skinny_swiglu_normed": [_p, _l, _p, _p, _l, _i, _i, _i, _f, _p, _l, C.POINTER(Plan), _p],
    "spacer_swiglu_f32_fwd": [_p, _p, _i, _i, _p],
    "spacer_gemm_swiglu_bf16": [_p, _l, _p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _p],
    "spacer_resize_bicubic_aa_u8": [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _i, _p, _p],
    "spacer_gather_frames_u8": [_p, _p, _p, _i, _l, _p],
    "spacer_sumsq_f32": [_p, _l, _p, _p],
    "spacer_adamw_step": [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _f, _p, _f, _f, _p],
    "spacer_split_f32_pair": [_p, _l, _p, _p, _l, _i, _i, _p],
    "spacer_act_f32_pair": [_p, _l, _p, _p, _l, _i, _i, _i, _p, _p],
    "spacer_swiglu_f32_pair": [_p, _p, _p, _i, _i, _p, _p],
    "spacer_norm_f32_pair": [_p, _p, _p, _p, _p, _i, _i, _f, _i, _p, _p, _p],
    "spacer_rope_f32_pair": [_p, _l, _p, _p, _p, _p, _l, _i, _i, _i, _i, _p],
    "spacer_embed_fwd_f32video": [_p, _p, _p, _p, _p, _i, _i, _p],
    "spacer_attn_fwd_pair": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _l, _l, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p],
}
OTHER_SYMBOLS = ["spacer_last_error", "spacer_version", "spacer_sample_workspace_bytes", "spacer_attn_decode_workspace_bytes", "spacer_gemm_tile", "spacer_gemm_workspace_bytes", "spacer_gemm_swiglu_fused", "spacer_gemm_pair_fused", "spacer_gemm_pair_epilogue_fused", "spacer_resize_workspace_bytes", "spacer_gemm_skinny_swiglu_workspace_bytes"]

_lib = None


def load() -> C.CDLL:
    """Load (once) and type the shared library.  Raises SpacerError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpacerError(
            f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()). "
            "spacer_amd has no CPU fallback for the hot path.")
    # torch first: its wheel bundles its own HIP runtime (libamdhip64) and the library must bind to THAT copy -- loaded the other way
    # round the process holds two runtimes and launches from this library fail with "no ROCm-capable device is detected"
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.spacer_last_error.restype = C.c_char_p
    lib.spacer_version.restype = C.c_int
    lib.spacer_sample_workspace_bytes.argtypes = [_i, _i]
    lib.spacer_sample_workspace_bytes.restype = C.c_long
    lib.spacer_gemm_tile.argtypes = [_i, _i, _i, _i, C.POINTER(Plan)]
    lib.spacer_gemm_tile.restype = _i
    lib.spacer_gemm_swiglu_fused.argtypes = [_i, _i, _i, C.POINTER(Plan)]
    lib.spacer_gemm_swiglu_fused.restype = _i
    lib.spacer_gemm_pair_fused.argtypes = [_i, _i, _i, _i, C.POINTER(Plan)]
    lib.spacer_gemm_pair_fused.restype = _i
    lib.spacer_gemm_pair_epilogue_fused.argtypes = [_i, _i, _i, _i, _i, _i, C.POINTER(Plan)]
    lib.spacer_gemm_pair_epilogue_fused.restype = _i
    lib.spacer_gemm_workspace_bytes.argtypes = []
    lib.spacer_gemm_workspace_bytes.restype = C.c_long
    lib.spacer_gemm_skinny_swiglu_workspace_bytes.argtypes = []
    lib.spacer_gemm_skinny_swiglu_workspace_bytes.restype = C.c_long
    lib.spacer_resize_workspace_bytes.argtypes = [_i, _i, _i]
    lib.spacer_resize_workspace_bytes.restype = C.c_long
    lib.spacer_attn_decode_workspace_bytes.argtypes = [_i, _i]
    lib.spacer_attn_decode_workspace_bytes.restype = C.c_long
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise SpacerError(f"{what} failed (code {rc}): {load().spacer_last_error().decode()}")
