"""ctypes binding of libullava_hip.so (the C-ABI declared in include/ullava_hip.h).

There is no fallback: if the library is missing or a symbol is absent this raises, and every op in
`ops.py` goes through here.  The library is built in-tree by `__graft_entry__.build()` /
`make -C u-llava_amd/csrc` so it travels with the repo snapshot.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ULL_LIB_PATH", os.path.join(_HERE, "csrc", "libullava_hip.so"))   # override: kernel A/B experiments only

_i64, _i32, _f32, _ptr = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p


# the structs of include/ullava_hip.h's coarse entries (one call = a whole stack of layers)
class Linear(ctypes.Structure):
    _fields_ = [("w", _ptr), ("w_tiled", _ptr), ("bias", _ptr), ("n", _i64), ("k", _i64), ("ldw", _i64)]


class LlamaLayer(ctypes.Structure):
    _fields_ = [("ln1", _ptr), ("ln2", _ptr), ("qkv", Linear), ("o", Linear), ("gu", Linear), ("down", Linear)]


class ClipLayer(ctypes.Structure):
    _fields_ = [("ln1_w", _ptr), ("ln1_b", _ptr), ("ln2_w", _ptr), ("ln2_b", _ptr), ("qkv", Linear), ("out", Linear), ("fc1", Linear), ("fc2", Linear)]


class SamBlock(ctypes.Structure):
    _fields_ = [("n1_w", _ptr), ("n1_b", _ptr), ("n2_w", _ptr), ("n2_b", _ptr), ("qkv", Linear), ("proj", Linear), ("lin1", Linear), ("lin2", Linear),
                ("rel_pos_h", _ptr), ("rel_pos_w", _ptr), ("window", _i64)]


# name -> argtypes (restype is int = ULL_OK / ULL_ERR_* unless listed in VALUE_RETURNING)
SIGNATURES = {
    "ull_gemm_bf16": [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _i32, _ptr, _i64, _ptr],
    "ull_gemm_streamk_ws_bytes": [],
    "ull_gemm_qkv_rope_bf16": [_ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _i32, _ptr, _i64, _ptr],
    "ull_rope_table_bf16": [_ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr],
    "ull_sam_self_attn_heads_bf16": [_ptr, _ptr, _i64, _i64, _i32] + [_ptr] * 7 + [_ptr],
    "ull_sam_out_ln_bf16": [_ptr, _i64, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _f32, _ptr, ctypes.POINTER(_ptr), ctypes.POINTER(_i32), _ptr],
    "ull_sam_token_mlp_ln_bf16": [_ptr, _ptr, _i64, _i64, _i64] + [_ptr] * 6 + [_f32, _ptr, _ptr, ctypes.POINTER(_ptr), ctypes.POINTER(_i32), _ptr],
    "ull_sam_small_mlps_bf16": [_ptr, _i64, _i64, ctypes.POINTER(_ptr), _i64, _i64, _i64, _ptr, _ptr, _ptr],
    "ull_sam_t2i_attention_bf16": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _ptr, _ptr],
    "ull_sam_i2t_attention_ln_bf16": [_ptr] * 4 + [_i64] * 3 + [_ptr] * 4 + [_i32, _ptr, _ptr, _f32, _ptr, _ptr],
    "ull_rmsnorm_bwd_bf16": [_ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _f32, _ptr],
    "ull_swiglu_fwd_bf16": [_ptr, _ptr, _i64, _i64, _i32, _ptr],
    "ull_swiglu_bwd_bf16": [_ptr, _ptr, _ptr, _i64, _i64, _i32, _ptr],
    "ull_relu_mask_bf16": [_ptr, _ptr, _ptr, _i64, _ptr],
    "ull_rope_bwd_inplace_bf16": [_ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "ull_attention_bwd_bf16": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.POINTER(_i64), _ptr, _i64, _i64, _i64, _i64, _i64, _i32,
                               _f32, _ptr, _ptr],
    "ull_attention_bwd_mfma_bf16": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, ctypes.POINTER(_i64), _ptr, _i64, _i64,
                                    _i64, _i64, _i64, _i32, _f32, _ptr, _ptr],
    "ull_shifted_cross_entropy_bwd_bf16": [_ptr, _i64, _ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr],
    "ull_embed_splice_bwd_bf16": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _i64, _i32, _ptr],
    "ull_colsum_bf16": [_ptr, _i64, _i64, _i64, _ptr, _ptr],
    "ull_transpose2d_bf16": [_ptr, _i64, _ptr, _i64, _i64, _i64, _ptr],
    "ull_sum_slabs_bf16": [_ptr, _ptr, _i64, _i64, _f32, _ptr],
    "ull_layernorm_bwd_bf16": [_ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _f32, _ptr],
    "ull_layernorm2d_cl_bwd_bf16": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _f32, _i32, _ptr],
    "ull_gelu_fwd_bf16": [_ptr, _ptr, _i64, _ptr],
    "ull_gelu_bwd_bf16": [_ptr, _ptr, _ptr, _i64, _ptr],
    "ull_mask_matmul_bwd_bf16": [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr],
    "ull_mask_loss_sums_bwd_f32": [_ptr, _ptr, _ptr, _i64, _i64, _f32, _ptr, _ptr],
    "ull_box_losses_bwd_f32": [_ptr, _i32, _ptr, _i64, _ptr, _ptr, _ptr],
    "ull_bilinear_bwd_f32": [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "ull_gemv_bf16": [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _i32, _ptr],
    "ull_dropout_apply_bf16": [_ptr, _ptr, _ptr, _i64, _f32, _ptr],
    "ull_gemm_skinny_bf16": [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _i32, _ptr],
    "ull_gemv_rmsnorm_bf16": [_ptr, _i64, _ptr, _f32, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _i32, _ptr],
    "ull_gemv_qkv_rope_append_bf16": [_ptr, _i64, _ptr, _f32, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                      _ptr],
    "ull_rmsnorm_bf16": [_ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _f32, _ptr],
    "ull_llama_prefill_layers_bf16": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _f32, _ptr,
                                      _i64, _i64, _ptr, _ptr],
    "ull_llama_decode_layers_bf16": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64,
                                     _i64, _i64, _f32, _ptr, _ptr],
    "ull_clip_layers_bf16": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _f32, _ptr, _i64, _i64, _ptr, _ptr],
    "ull_sam_blocks_bf16": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _f32, _ptr, _i64, _i64, _ptr, _ptr],
    "ull_shifted_cross_entropy_bf16": [_ptr, _i64, _ptr, _i64, _i64, _i64, _ptr, _ptr],
    "ull_layernorm_bf16": [_ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _f32, _ptr],
    "ull_clip_embed_ln_bf16": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _f32, _ptr],
    "ull_attention_bf16": [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _i64,
                           _ptr, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _f32, _f32, _ptr, _ptr, _i64, _i64, _i32, _ptr, _ptr],
    "ull_rope_inplace_bf16": [_ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "ull_rope_append_bf16": [_ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _ptr],
    "ull_transpose_v_bf16": [_ptr, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr],
    "ull_sam_window_attention_bf16": [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _ptr, _ptr],
    "ull_interp_rows_linear_bf16": [_ptr, _ptr, _i64, _i64, _i64, _ptr],
    "ull_im2col_bf16": [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "ull_mm_spans": [_ptr, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr],
    "ull_greedy_step_bf16": [_ptr, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _i32, _ptr, _i64, _i64, _ptr, _ptr],
    "ull_embed_splice_bf16": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr],
    "ull_video_pool_bf16": [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _i64, _ptr],
    "ull_gather_rows_bf16": [_ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "ull_add_rows_bf16": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _ptr],
    "ull_window_partition_bf16": [_ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr],
    "ull_window_unpartition_add_bf16": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr],
    "ull_sam_relpos_bf16": [_ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _i64, _ptr],
    "ull_layernorm2d_cl_bf16": [_ptr, _ptr, _ptr, _ptr, _i64, _i64, _f32, _i32, _ptr],
    "ull_im2col3x3_bf16": [_ptr, _ptr, _i64, _i64, _i64, _i64, _ptr],
    "ull_mask_matmul_bf16": [_ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr],
    "ull_bilinear_f32": [_ptr, _i32, _i64, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _ptr],
    "ull_patchify_bf16": [_ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _ptr, _ptr],
    "ull_resample_u8": [_ptr, _i64, _i64, _i64, _i32, _i64, _ptr, _ptr, _i64, _ptr, _ptr],
    "ull_u8_lut_chw": [_ptr, _i64, _i64, _i64, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _i64, _i32, _ptr],
    "ull_mask_iou_counts": [_ptr, _ptr, _i64, _i64, _i32, _ptr, _ptr],
    "ull_mask_loss_sums_f32": [_ptr, _ptr, _i64, _i64, _f32, _ptr, _ptr],
    "ull_box_losses_f32": [_ptr, _i32, _ptr, _i64, _ptr, _ptr],
    "ull_adamw_step_f32": [_ptr, _ptr, _ptr, _ptr, _i32, _ptr, _i32, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _ptr],
    "ull_sumsq_f32": [_ptr, _i32, _i64, _ptr, _ptr],
}

# every dtype-dependent entry point exists twice: ull_*_bf16 (bfloat16 build) and ull_*_f16 (IEEE binary16 build), same signature
SIGNATURES.update({name[:-4] + "f16": args for name, args in list(SIGNATURES.items()) if name.endswith("_bf16")})

# fp32 build of the inference path (csrc/f32.hip): the same signatures as the bf16 entries of the same name
F32_TWINS = ['ull_gemm', 'ull_attention', 'ull_transpose_v', 'ull_rmsnorm', 'ull_layernorm', 'ull_clip_embed_ln', 'ull_layernorm2d_cl', 'ull_rope_inplace', 'ull_rope_append', 'ull_im2col', 'ull_im2col3x3', 'ull_video_pool', 'ull_add_rows', 'ull_window_unpartition_add', 'ull_sam_relpos', 'ull_interp_rows_linear', 'ull_mask_matmul', 'ull_greedy_step', 'ull_shifted_cross_entropy']
SIGNATURES.update({name + "_f32": SIGNATURES[name + "_bf16"] for name in F32_TWINS})

# fp16-only entry points (no bf16 twin): the fp32 neck of an fp16 SAM encoder (image_encoder.py:117-124)
SIGNATURES["ull_neck_layernorm2d_f32in_f16"] = [_ptr, _ptr, _f32, _ptr, _ptr, _ptr, _ptr, _i64, _i64, _f32, _ptr]

VALUE_RETURNING = {"ull_gemm_streamk_ws_bytes": _i64}      # plain queries: the return value is the answer, not a status

ERRORS = {-1: "ULL_ERR_ARG (null pointer / bad size)", -2: "ULL_ERR_SHAPE (alignment or shape constraint)",
          -3: "ULL_ERR_LAUNCH (HIP launch failed)", -4: "ULL_ERR_LDS (does not fit the LDS budget)"}

_lib = None


def load():
    """Load the shared library (once) and attach argtypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"u-llava_amd: HIP library not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C u-llava_amd/csrc`. There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = VALUE_RETURNING.get(name, ctypes.c_int)
    _lib = lib
    return lib


def call(name: str, *args):
    fn = getattr(load(), name, None)
    if fn is None or name not in SIGNATURES:
        # (an op whose dtype has no kernel build: e.g. a backward / fused entry asked for float32 -- the fp32 build covers the inference path only)
        raise RuntimeError(f"u-llava_amd: the library has no entry point `{name}` (no kernel build of this op for that element type)")
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"u-llava_amd: {name} failed: {ERRORS.get(rc, rc)}")


def query(name: str, *args):
    """A VALUE_RETURNING entry point: returns its value."""
    fn = getattr(load(), name, None)
    if fn is None or name not in SIGNATURES:
        raise RuntimeError(f"u-llava_amd: the library has no entry point `{name}` (no kernel build of this op for that element type)")
    return fn(*args)
