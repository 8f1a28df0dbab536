"""ctypes binding of libboxdreamer_hip.so (the C ABI in include/boxdreamer_hip.h).

There is NO fallback: if the library is missing or a call fails, this raises.  PyTorch is used
only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BOXDREAMER_HIP_LIB") or os.path.join(HERE, "libboxdreamer_hip.so")   # env: deploy / A-B a build

DTYPE_BF16, DTYPE_F16, DTYPE_F32 = 0, 1, 2
PREC_BF16, PREC_F16, PREC_BF16X3, PREC_F16_OUT_BF16X3, PREC_FP8, PREC_BF16_OUT_FP8 = 0, 1, 2, 3, 4, 5
PREC_BF16X3_ATTN_X3 = 6                                # whole-path only: split-bf16 attention everywhere (7, 11, 12: removed in ABI 7)
PREC_F16C8 = 8                                         # f16 + e4m3 corrections (include/boxdreamer_hip.h)
PREC_F16_OUT_F16C8, PREC_BF16X3_OUT_F16C8 = 9, 10      # bd_attention[_q] only: f16 / split-bf16 attention, F16C8 operand out
PREC_F16C8_QK16 = 13                                   # whole-path only: F16C8 Linears, BETR's QKV split: q, k one f16 pass, v F16C8
PREC_F16X3, PREC_F16X3_ATTN_X3 = 14, 15                # split-f16 Linears (the promoted class); _ATTN_X3: split-bf16 attention everywhere
PREC_F16_OUT_F16X3, PREC_BF16X3_OUT_F16X3 = 16, 17     # bd_attention[_q] only: split-f16 planes out
F16C8_D = 11                                           # lo planes are scaled 2^D above their q plane
PREC_NAMES = {"bf16": PREC_BF16, "fp16": PREC_F16, "f16": PREC_F16, "bf16x3": PREC_BF16X3, "fp8": PREC_FP8,
              "bf16x3_attn_x3": PREC_BF16X3_ATTN_X3, "f16c8": PREC_F16C8, "f16c8_qk16": PREC_F16C8_QK16,
              "f16x3": PREC_F16X3, "f16x3_attn_x3": PREC_F16X3_ATTN_X3}
_F16X3_FAMILY = (PREC_F16X3, PREC_F16X3_ATTN_X3)
# "fp8_mixed" (configs[4], usable form): the e4m3 class with the precision-critical Linears kept in bf16 through the per-Linear
# promotion bits -- a POLICY over BD_PREC_FP8, not another library mode (fp8_mixed_policy below; the modules apply it at construction)
PREC_NAMES["fp8_mixed"] = PREC_FP8


def promoted_class(base: int) -> int:
    """Operand class a promoted Linear runs in (include/boxdreamer_hip.h: BD_PROMOTE_*)."""
    return PREC_F16X3 if base == PREC_F16C8 else (PREC_BF16 if base == PREC_FP8 else base)


def fp8_mixed_policy(depth: int, normed: bool):
    """(per-block masks, misc mask) of the mixed e4m3 mode.  e4m3 (3 mantissa bits) where the consumer is forgiving -- the MLPs and
    DINOv2's QKV (2/3 of the Linear FLOPs); bf16 where a rounding lands on the residual stream or the heatmap un-damped: every proj,
    BETR's QKV (its v columns decide the block's output; q, k are RMS-normalised from the rounded values), the adapter and the head
    (per-Linear sensitivities: profiles/r3_strict_modes.md section 1, profiles/r4_fp8_mixed.md)."""
    if normed:      # BETR
        return [PROMOTE_QKV | PROMOTE_PROJ] * depth, PROMOTE_ADAPTER_FC1 | PROMOTE_ADAPTER_FC2 | PROMOTE_BBOX_PROJ
    return [PROMOTE_PROJ] * depth, 0
_X3_FAMILY = (PREC_BF16X3, PREC_BF16X3_ATTN_X3)
ACT_NONE, ACT_GELU = 0, 1
# per-Linear promotion: F16C8 family -> split-f16, e4m3 -> bf16 (include/boxdreamer_hip.h: BD_PROMOTE_*)
PROMOTE_QKV, PROMOTE_PROJ, PROMOTE_FC1, PROMOTE_FC2, PROMOTE_ATTN = 1, 2, 4, 8, 16
PROMOTE_ADAPTER_FC1, PROMOTE_ADAPTER_FC2, PROMOTE_BBOX_EMB, PROMOTE_BBOX_PROJ = 1, 2, 4, 8
PROMOTE_PATCH_EMBED = 1
# The precision a module runs when its config names none: the fastest mode that MEETS the path's parity bar (heatmap logits
# within 1e-3 of the fp32 CPU forward, identical top-20 sets).  "bf16" -- the reference's own `precision`, 4e-2 off its fp32
# forward -- is the explicit throughput opt-in (`hip_precision: bf16` in the decoder / encoder config, or $BOXDREAMER_HIP_PREC).
DEFAULT_PREC = "f16c8_qk16"

_ERR = {-1: "BD_ERR_SHAPE", -2: "BD_ERR_DTYPE", -3: "BD_ERR_ALIGN", -4: "BD_ERR_WORKSPACE", -5: "BD_ERR_NULL"}


class HipLibraryError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("a_plane", C.c_int64),
                ("W", C.c_void_p), ("ldw", C.c_int64), ("w_plane", C.c_int64),
                ("bias", C.c_void_p), ("wscale", C.c_void_p),
                ("resid", C.c_void_p), ("ldr", C.c_int64),
                ("addtab", C.c_void_p), ("tab_rows", C.c_int),
                ("out", C.c_void_p), ("ldo", C.c_int64), ("out_plane", C.c_int64),
                ("out_f32", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("act", C.c_int),
                ("rpg_in", C.c_int), ("rpg_out", C.c_int), ("row_off", C.c_int), ("w_qexp", C.c_int),
                ("rms_wq", C.c_void_p), ("rms_wk", C.c_void_p), ("rms_eps", C.c_float), ("rms_parts", C.c_int),
                ("ln_stats_out", C.c_void_p), ("ln_op_out", C.c_void_p), ("ln_op_plane", C.c_int64), ("ln_op_ld", C.c_int64),
                ("ln_stats_in", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_eps", C.c_float), ("ln_resid_in_op", C.c_int),
                ("sk_ws", C.c_void_p), ("sk_split", C.c_int)]


class Linear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("wscale", C.c_void_p), ("w_qexp", C.c_int)]


class BlockWeights(C.Structure):
    _fields_ = [("ln1_w", C.c_void_p), ("ln1_b", C.c_void_p), ("ln2_w", C.c_void_p), ("ln2_b", C.c_void_p),
                ("qkv", Linear), ("proj", Linear), ("fc1", Linear), ("fc2", Linear),
                ("q_norm_w", C.c_void_p), ("k_norm_w", C.c_void_p), ("qkv16", Linear), ("promote", C.c_int),
                ("qkv_f", Linear), ("fc1_f", Linear), ("qkv16_f", Linear),
                ("qkv_s", C.c_void_p), ("fc1_s", C.c_void_p), ("qkv16_s", C.c_void_p), ("ln_resid3", C.c_int)]


class DinoWeights(C.Structure):
    _fields_ = [("depth", C.c_int), ("dim", C.c_int), ("heads", C.c_int), ("n_prefix", C.c_int),
                ("grid", C.c_int), ("patch", C.c_int), ("kpad", C.c_int),
                ("ln_eps", C.c_float),
                ("patch_embed", Linear),
                ("pos_patch", C.c_void_p), ("prefix_tokens", C.c_void_p),
                ("norm_w", C.c_void_p), ("norm_b", C.c_void_p),
                ("blocks", C.POINTER(BlockWeights)), ("promote_misc", C.c_int), ("feats_prec", C.c_int), ("latency_mode", C.c_int)]


class BetrWeights(C.Structure):
    _fields_ = [("depth", C.c_int), ("dim", C.c_int), ("heads", C.c_int), ("grid", C.c_int),
                ("patch", C.c_int), ("box_dim", C.c_int), ("kpad", C.c_int),
                ("ln_eps", C.c_float), ("adapter_ln_eps", C.c_float), ("rms_eps", C.c_float),
                ("adapter_fc1", Linear), ("adapter_fc2", Linear), ("bbox_emb", Linear), ("bbox_proj", Linear),
                ("pos_table", C.c_void_p), ("query_token", C.c_void_p),
                ("blocks", C.POINTER(BlockWeights)), ("promote_misc", C.c_int), ("latency_mode", C.c_int)]


class TraceRecord(C.Structure):
    _fields_ = [("kind", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("ms", C.c_float)]


EXPORTS = [
    "bd_abi_version", "bd_target_arch", "bd_gemm", "bd_layernorm", "bd_qk_rmsnorm", "bd_attention",
    "bd_im2col_images", "bd_patchify_heatmaps", "bd_write_prefix_tokens", "bd_query_substitute",
    "bd_gather_query_tokens", "bd_unpatchify_sigmoid", "bd_decode_topk",
    "bd_encoder_workspace_bytes", "bd_encoder_forward", "bd_decoder_workspace_bytes", "bd_decoder_forward",
    "bd_trace_begin", "bd_trace_end", "bd_render_corner_heatmaps", "bd_attention_q", "bd_gather_query_rows_f32",
    "bd_dino_match_scores", "bd_topk_mask", "bd_solve_pnp", "bd_gemm_fuses_qk_rmsnorm", "bd_gemm_takes_ln_fold", "bd_gemm_splitk_flag_bytes",
    "bd_gemm_splitk_workspace_bytes", "bd_solve_pnp_host", "bd_attention_prefix",
    "bd_lanes_prepare", "bd_encoder_workspace_bytes_lanes", "bd_encoder_forward_lanes", "bd_decoder_workspace_bytes_lanes",
    "bd_decoder_forward_lanes",
]

_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raise loudly if it has not been built."""
    global _lib
    # every wrapper calls load() before it hands the first tensor to ptr(): forget a device noted by a call that raised
    # between ptr() and stream() (it would make the next call on another device fail with a false "mixes tensors" error)
    _call.dev = None
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: the gfx950 HIP kernels are not built. Run "
            "`python -m boxdreamer_amd.build` (or __graft_entry__.build()). There is no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise HipLibraryError(f"{LIB_PATH} does not export {name}")
    i, i64, vp, f, sz = C.c_int, C.c_int64, C.c_void_p, C.c_float, C.c_size_t
    lib.bd_abi_version.restype = i
    lib.bd_target_arch.restype = C.c_char_p
    lib.bd_gemm.argtypes = [C.POINTER(GemmArgs), i, vp]
    lib.bd_gemm_fuses_qk_rmsnorm.argtypes = [C.POINTER(GemmArgs), i]
    lib.bd_gemm_takes_ln_fold.argtypes = [C.POINTER(GemmArgs), i]
    lib.bd_gemm_splitk_flag_bytes.argtypes = [i, i]
    lib.bd_gemm_splitk_flag_bytes.restype = sz
    lib.bd_gemm_splitk_workspace_bytes.argtypes = [i, i]
    lib.bd_gemm_splitk_workspace_bytes.restype = sz
    lib.bd_layernorm.argtypes = [vp, i64, vp, vp, f, vp, i64, vp, i64, i, i, i, i, i, i, vp]
    lib.bd_qk_rmsnorm.argtypes = [vp, i64, vp, vp, f, i, i, i, i, vp]
    lib.bd_attention.argtypes = [vp, i64, vp, i64, i, i, i, i, f, i, vp]
    lib.bd_im2col_images.argtypes = [vp, i, vp, i64, i, i, i, i, i, vp]
    lib.bd_patchify_heatmaps.argtypes = [vp, i, vp, i64, i, i, i, i, i, i, vp]
    lib.bd_write_prefix_tokens.argtypes = [vp, vp, i, i, i, i, vp]
    lib.bd_query_substitute.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp]
    lib.bd_gather_query_tokens.argtypes = [vp, vp, vp, i64, i, i, i, i, i, vp]
    lib.bd_unpatchify_sigmoid.argtypes = [vp, vp, vp, i, i, i, i, vp]
    lib.bd_decode_topk.argtypes = [vp, i, i, i, i, vp, vp, vp, vp]
    lib.bd_encoder_workspace_bytes.argtypes = [C.POINTER(DinoWeights), i, i]
    lib.bd_encoder_workspace_bytes.restype = sz
    lib.bd_encoder_forward.argtypes = [C.POINTER(DinoWeights), vp, i, i, i, vp, vp, i64, vp, sz, i, vp]
    lib.bd_decoder_workspace_bytes.argtypes = [C.POINTER(BetrWeights), i, i, i]
    lib.bd_decoder_workspace_bytes.restype = sz
    lib.bd_decoder_forward.argtypes = [C.POINTER(BetrWeights), vp, i, vp, i64, vp, i, i, i, vp, vp, vp, sz, i, vp]
    lib.bd_encoder_workspace_bytes_lanes.argtypes = [C.POINTER(DinoWeights), i, i, i]
    lib.bd_encoder_workspace_bytes_lanes.restype = sz
    lib.bd_encoder_forward_lanes.argtypes = [C.POINTER(DinoWeights), vp, i, i, i, vp, vp, i64, vp, sz, i, i, vp]
    lib.bd_decoder_workspace_bytes_lanes.argtypes = [C.POINTER(BetrWeights), i, i, i, i]
    lib.bd_decoder_workspace_bytes_lanes.restype = sz
    lib.bd_decoder_forward_lanes.argtypes = [C.POINTER(BetrWeights), vp, i, vp, i64, vp, i, i, i, vp, vp, vp, sz, i, i, vp]
    lib.bd_lanes_prepare.argtypes = []
    lib.bd_attention_q.argtypes = [vp, i64, vp, i64, i, i, i, i, f, vp, i, i, vp]
    lib.bd_attention_prefix.argtypes = [vp, i64, vp, i64, i, i, i, i, f, i, i, i, vp]
    lib.bd_gather_query_rows_f32.argtypes = [vp, vp, vp, i, i, i, i, vp]
    lib.bd_render_corner_heatmaps.argtypes = [vp, i, i, i, i, vp, i, vp]
    lib.bd_dino_match_scores.argtypes = [vp, vp, i, vp, i, i, i, i, i, i, f, vp, vp, vp, vp]
    lib.bd_topk_mask.argtypes = [vp, i, i, i, vp, vp]
    lib.bd_solve_pnp.argtypes = [vp, vp, vp, i, i, i, vp, vp]
    lib.bd_solve_pnp_host.argtypes = [vp, vp, vp, i, i, i, vp, i]
    lib.bd_trace_begin.argtypes = [i]
    lib.bd_trace_end.argtypes = [C.POINTER(TraceRecord), i]
    if lib.bd_abi_version() != 9:
        raise HipLibraryError("libboxdreamer_hip.so ABI version mismatch")
    _lib = lib
    return lib


# Device discipline, kept central so that no call site can forget it: `ptr()` notes the device of every tensor whose address
# is handed to the library, `stream()` returns torch's current stream ON THAT DEVICE and makes the device current for the
# launch (a module moved with .to("cuda:1") while cuda:0 is current would otherwise launch on a device-0 stream with
# device-1 pointers), and `check()` restores the previous current device.  Mixed devices in one call raise.
_call = threading.local()


def check(rc: int, what: str) -> None:
    prev = getattr(_call, "restore", None)
    _call.dev = None
    if prev is not None:
        _call.restore = None
        torch.cuda.set_device(prev)
    if rc == 0:
        return
    if rc < 0:
        raise HipLibraryError(f"{what} rejected its arguments: {_ERR.get(rc, rc)}")
    raise HipLibraryError(f"{what} failed with hipError_t {rc}")


def prec_id(prec) -> int:
    if isinstance(prec, int):
        return prec
    try:
        return PREC_NAMES[prec]
    except KeyError:
        raise ValueError(f"unknown precision {prec!r}; choose from {sorted(PREC_NAMES)}") from None


def operand_prec(prec) -> int:
    """Operand class of a (possibly whole-path) precision id: the value the unit operators and the weight packer take."""
    pid = prec_id(prec)
    if pid == PREC_F16C8_QK16:
        return PREC_F16C8
    if pid in _F16X3_FAMILY:
        return PREC_F16X3
    return PREC_BF16X3 if pid in _X3_FAMILY else pid


def op_dtype(prec) -> torch.dtype:
    pid = prec_id(prec)
    if pid == PREC_FP8:
        return torch.float8_e4m3fn          # OCP e4m3 (gfx950), not MI300's fnuz
    return torch.float16 if pid in (PREC_F16, PREC_F16C8, PREC_F16C8_QK16) + _F16X3_FAMILY else torch.bfloat16     # F16C8: plane 1 holds raw e4m3 bytes


def k_multiple(prec) -> int:
    """K padding granularity of GEMM operands: one 128-byte tile row per slab."""
    return 128 if prec_id(prec) == PREC_FP8 else 64


def planes(prec) -> int:
    return 2 if prec_id(prec) in _X3_FAMILY + _F16X3_FAMILY or prec_id(prec) in (PREC_F16C8, PREC_F16C8_QK16) else 1


AUTO_LANES_MIN_VIEWS = 24      # one batch runs as two sub-batch lanes from this many (sample, view) images on (profiles/r4_subbatch_lanes.md)
# ... earlier in the classes whose small launches were re-measured in round 5 (profiles/r5_small_experiments.md: lanes at batch 2 / 3):
# bf16 / f16 from batch 2 at T = 6 (-7.5 % / -4 % per step), the F16C8 class from batch 3 (-6 %; at batch 2 two lanes cost 9 %)
AUTO_LANES_MIN_VIEWS_BY_CLASS = {PREC_BF16: 12, PREC_F16: 12, PREC_F16C8: 18}


def resolve_lanes(setting, views: int, samples: int, prec=None) -> int:
    """Sub-batch lanes of one whole-path call (include/boxdreamer_hip.h, ABI v6+): `setting` is "auto" or 1..4; `views` = images of
    the call (B x T), `samples` = the units the batch can be cut at.  Bit-identical results for every value.  "auto": two lanes from
    AUTO_LANES_MIN_VIEWS images on (per class: AUTO_LANES_MIN_VIEWS_BY_CLASS), except in the e4m3 class, whose half-batch GEMMs lose more than the filled tail rounds win
    (measured: -2.7 % at batch 64, -4 % at batch 32; 16-bit classes +2 ... +9 %)."""
    if setting in (None, "auto"):
        cls = operand_prec(prec) if prec is not None else None
        n = 2 if views >= AUTO_LANES_MIN_VIEWS_BY_CLASS.get(cls, AUTO_LANES_MIN_VIEWS) and cls != PREC_FP8 else 1
    else:
        n = int(setting)
        if not 1 <= n <= 4:
            raise ValueError(f"hip_lanes must be 'auto' or 1..4, got {setting!r}")
    return max(1, min(n, samples))


def dtype_id(t: torch.Tensor) -> int:
    try:
        return {torch.bfloat16: DTYPE_BF16, torch.float16: DTYPE_F16, torch.float32: DTYPE_F32}[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported input dtype {t.dtype} (bf16 / fp16 / fp32)") from None


def ptr(t) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise HipLibraryError("the HIP path needs device tensors (got a CPU tensor); there is no CPU fallback")
    dev = getattr(_call, "dev", None)
    if dev is None:
        _call.dev = t.device
    elif dev != t.device:
        _call.dev = None
        raise HipLibraryError(f"one call mixes tensors on {dev} and {t.device}")
    return C.c_void_p(t.data_ptr())


def stream() -> C.c_void_p:
    """torch's current HIP stream on the device of the tensors passed through ptr() for this call; that device is made
    current until check() runs."""
    dev = getattr(_call, "dev", None)
    _call.dev = None
    if dev is None:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cur = torch.cuda.current_device()
    if dev.index is not None and dev.index != cur:
        _call.restore = cur
        torch.cuda.set_device(dev)
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def same_device(*tensors) -> torch.device:
    """All given (non-None) tensors must live on one HIP device; returns it."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipLibraryError("the HIP path needs device tensors (got a CPU tensor); there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise HipLibraryError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise HipLibraryError("no HIP device visible: BoxDreamer's MI355X path cannot run (no CPU fallback)")
