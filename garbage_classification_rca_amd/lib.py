"""ctypes binding of libmmrca.so (the C ABI declared in include/mmrca.h).

The product path has no CPU fallback: if the shared library is missing or a call fails this module raises.
Build with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C garbage_classification_rca_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmmrca.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_GELU_BWD, ACT_GELU_SAVE_GRAD, ACT_MUL, ACT_GELU_SAVE_GRAD_BF16 = 0, 1, 2, 3, 4, 5
ROWK, KROW = 0, 1
IMPL_AUTO, IMPL_REF, IMPL_MFMA, IMPL_MFMA256, IMPL_MFMA_PERSIST, IMPL_MFMA_BK32, IMPL_MFMA_1STAGE, IMPL_MFMA_TALL, IMPL_MFMA_256W, IMPL_MFMA_256X4 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9

HEAD_FIELDS = [f"{blk}_{leaf}" for blk in ("sai", "sat", "c1", "c2")
               for leaf in ("wq", "bq", "wk", "bk", "wv", "bv", "g", "b")] + ["fin_w", "fin_b"]


class HeadPtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in HEAD_FIELDS]


class MmrcaError(RuntimeError):
    pass


def build_library(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into libmmrca.so (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", src_dir, "-j8"]
    if force:
        subprocess.run(["make", "-C", src_dir, "clean"], check=True, capture_output=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise MmrcaError("building libmmrca.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


_lib = None

_i64, _i32, _f32, _vp, _u64 = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_uint64
_SIGS = {
    "mmrca_gemm": [_vp] * 6 + [_i64] * 6 + [_i32] * 6 + [_vp],
    "mmrca_gemm_colsum": [_vp] * 7 + [_i64] * 6 + [_i32] * 5 + [_vp],
    "mmrca_gemm_rows": [_vp] * 7 + [_i64] * 8 + [_i32] * 5 + [_vp],
    "mmrca_gemm_splitk": [_vp] * 4 + [_i64] * 7 + [_i32] * 2 + [_vp],
    "mmrca_gemm_streamk_workspace": [_vp, _i64, _vp],
    "mmrca_gemm_streamk_config": [_i32, _i32, _i32],
    "mmrca_gemm_x3": [_vp] * 10 + [_i64] * 6 + [_i32] * 5 + [_vp],
    "mmrca_gemm_splitk_x3": [_vp] * 6 + [_i64] * 7 + [_i32] * 2 + [_vp],
    "mmrca_split_f32": [_vp, _vp, _vp, _i64, _vp],
    "mmrca_sgd_step_x3": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _vp],
    "mmrca_adamw_step_x3": [_vp] * 6 + [_i64] + [_f32] * 5 + [_i32, _f32, _vp],
    "mmrca_nchw_to_rows": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "mmrca_im2row3x3": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _vp],
    "mmrca_col2im3x3": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _vp],
    "mmrca_im2row3x3_tap": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _vp],
    "mmrca_col2im3x3_tap": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _vp],
    "mmrca_dwconv3x3_fwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "mmrca_dwconv3x3_bwd_ws": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp],
    "mmrca_dwconv3x3_bwd": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "mmrca_conv3x3_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "mmrca_conv_bn_finish": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _vp],
    "mmrca_conv3x3_wgrad": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "mmrca_gemm_bnstats": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp],
    "mmrca_bn_finish_sums": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _vp],
    "mmrca_bn_act_bwd_sums": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "mmrca_se_dx": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp],
    "mmrca_bn_act_fwd_res": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i32, _vp],
    "mmrca_bn_moments": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp],
    "mmrca_bn_act_fwd_fin": [_vp] * 13 + [_i64, _i32, _i32, _i64, _f32, _f32, _i32, _vp],
    "mmrca_bn_stats": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _f32, _f32, _i32, _i32, _vp],
    "mmrca_bn_stats_ws": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _f32, _f32, _i32, _i32, _vp, _i64, _i32, _vp],
    "mmrca_bn_stats_fused": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _f32, _f32, _i32, _vp, _i32, _vp],
    "mmrca_bn_act_bwd_ws": [_vp] * 10 + [_i64, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _vp],
    "mmrca_bn_act_fwd": [_vp] * 6 + [_i64, _i32, _i32, _i32, _vp],
    "mmrca_bn_act_bwd": [_vp] * 10 + [_i64, _i32, _i32, _i32, _i32, _vp],
    "mmrca_rowpool_mean": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "mmrca_rowpool_mean_bwd": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "mmrca_se_scale_fwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "mmrca_se_mlp_fwd": [_vp] * 9 + [_i32] * 4 + [_vp],
    "mmrca_se_mlp_bwd": [_vp] * 16 + [_i32] * 4 + [_vp],
    "mmrca_se_scale_bwd": [_vp] * 5 + [_i32, _i32, _i32, _i32, _vp],
    "mmrca_bias_act_fwd": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "mmrca_bias_act_bwd": [_vp] * 5 + [_i64, _i32, _i32, _i32, _vp],
    "mmrca_residual_add": [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp],
    "mmrca_maxpool3x3s2_fwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "mmrca_maxpool3x3s2_bwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "mmrca_channel_gather": [_vp, _vp, _vp, _i64, _i32, _i32, _i64, _i32, _i32, _vp],
    "mmrca_channel_interleave2": [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp],
    "mmrca_channel_deinterleave2": [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp],
    "mmrca_image_preprocess": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "mmrca_image_rotate_crop": [_vp, _vp, _i32, _i32, _vp],
    "mmrca_image_resize_u8": [_vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "mmrca_image_augment": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "mmrca_colsum_accum": [_vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "mmrca_gelu_bwd": [_vp, _vp, _vp, _i64, _i32, _vp],
    "mmrca_gelu_bwd_colsum": [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "mmrca_mha_fwd": [_vp] * 4 + [_i32] * 4 + [_f32, _f32, _u64, _vp, _i32, _i32, _vp],
    "mmrca_mha_fwd_planes": [_vp] * 6 + [_i32] * 4 + [_f32, _f32, _u64, _vp, _vp],
    "mmrca_mha_fwd_planes_in": [_vp] * 7 + [_i32] * 4 + [_f32, _f32, _u64, _vp, _vp],
    "mmrca_mha_fwd_x3": [_vp] * 6 + [_i32] * 4 + [_f32, _f32, _u64, _vp, _vp],
    "mmrca_mha_bwd": [_vp] * 6 + [_i32] * 4 + [_f32, _f32, _u64, _vp, _i32, _i32, _vp],
    "mmrca_mha_bwd_colsum": [_vp] * 7 + [_i64] + [_i32] * 4 + [_f32, _f32, _u64, _vp, _i32, _i32, _vp],
    "mmrca_mha_cross_fwd": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64] + [_i32] * 5 + [_f32, _f32, _u64, _i32, _i32, _vp],
    "mmrca_mha_cross_fwd_x3": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64] + [_i32] * 5 + [_f32, _f32, _u64, _vp],
    "mmrca_mha_cls_fwd": [_vp] * 4 + [_i32] * 4 + [_f32, _f32, _u64, _vp, _i32, _vp],
    "mmrca_mha_cls_bwd": [_vp] * 6 + [_i32] * 4 + [_f32, _f32, _u64, _vp, _i32, _vp],
    "mmrca_add_layernorm_fwd": [_vp] * 8 + [_i64, _i32, _i64, _i64, _f32, _f32, _u64, _f32, _u64, _i32, _vp],
    "mmrca_add_layernorm_fwd_x3": [_vp] * 10 + [_i64, _i32, _i64, _i64, _f32, _f32, _u64, _f32, _u64, _vp],
    "mmrca_layernorm_bwd": [_vp] * 9 + [_i64, _i32, _i64, _i64, _i64, _f32, _u64, _f32, _u64, _vp, _vp, _vp, _i32, _vp],
    "mmrca_layernorm_bwd_mixed": [_vp] * 9 + [_i64, _i32, _i64, _i64, _i64, _f32, _u64, _f32, _u64, _vp, _vp, _vp, _vp],
    "mmrca_embed_fwd": [_vp] * 6 + [_i64, _i32, _i32, _vp],
    "mmrca_embed_bwd": [_vp] * 6 + [_i64, _i32, _i32, _i32, _i32, _vp],
    "mmrca_patchify_fwd": [_vp, _vp] + [_i32] * 6 + [_vp],
    "mmrca_vit_assemble_fwd": [_vp] * 4 + [_i32] * 4 + [_vp],
    "mmrca_vit_assemble_bwd": [_vp] * 4 + [_i32] * 4 + [_vp],
    "mmrca_head_fwd": [_vp, _vp, C.POINTER(HeadPtrs), _vp] + [_i32] * 6 + [_f32, _u64, _i32, _vp],
    "mmrca_head_bwd": [_vp, _vp, _vp, C.POINTER(HeadPtrs), C.POINTER(HeadPtrs), _vp, _vp] + [_i32] * 6 + [_f32, _u64, _i32, _vp, _i64, _vp],
    "mmrca_xent_fwd_bwd": [_vp, _vp, _vp, _f32, _vp, _vp, _i32, _i32, _f32, _vp],
    "mmrca_sgd_step": [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _vp],
    "mmrca_adamw_step": [_vp] * 5 + [_i64] + [_f32] * 5 + [_i32, _f32, _vp],
    "mmrca_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "mmrca_seed_epoch_set": [_vp, _u64, _vp],
    "mmrca_sd_rowscale": [_vp, _vp, _i32, _i32, _u64, _vp],
}
EXPORTS = sorted(list(_SIGS) + ["mmrca_last_error", "mmrca_version", "mmrca_debug_set", "mmrca_bn_flat_set", "mmrca_debug_attn_stamps", "mmrca_gemm_splitk_workspace_bytes", "mmrca_gemm_streamk_workspace_bytes",
                                  "mmrca_head_bwd_workspace_bytes", "mmrca_conv3x3_stat_slots", "mmrca_gemm_bnstats_slots"])


def load(build_if_missing: bool = False):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            build_library()
        else:
            raise MmrcaError(f"{LIB_PATH} not found: the HIP extension is required (run __graft_entry__.build()); "
                             "there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.mmrca_gemm_splitk_workspace_bytes.argtypes = [_i64, _i64]
    lib.mmrca_gemm_splitk_workspace_bytes.restype = _i64
    lib.mmrca_gemm_streamk_workspace_bytes.argtypes = []
    lib.mmrca_gemm_streamk_workspace_bytes.restype = _i64
    lib.mmrca_conv3x3_stat_slots.argtypes = [_i32, _i32, _i32]
    lib.mmrca_conv3x3_stat_slots.restype = _i64
    lib.mmrca_last_error.restype = C.c_char_p
    lib.mmrca_version.restype = C.c_int
    _lib = lib
    if os.environ.get("MMRCA_DBG"):            # kernel ablation / probe bits (csrc: g_mmrca_dbg); never set in production runs
        lib.mmrca_debug_set(int(os.environ["MMRCA_DBG"]))
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        raise MmrcaError(f"{what} failed ({rc}): {load().mmrca_last_error().decode()}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise MmrcaError(f"unsupported dtype {dt}")


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """the current HIP stream of the current device as an integer handle.  Called once per launch (~500 per step): the raw C
    accessors skip the torch.cuda.Stream object that torch.cuda.current_stream() builds (1.5 us each)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise MmrcaError(f"{name}: tensor must live in HBM (got device {t.device}); the product path has no CPU fallback")


# ---------------------------------------------------------------------------------------------------------
# thin typed wrappers (raw pointers in, nothing allocated)
# ---------------------------------------------------------------------------------------------------------
GEMM_PROFILE = None     # bench.py sets this to a list: every MFMA-qualified GEMM launch is bracketed by HIP events
KERNEL_PROFILE = None   # bench.py sets this to a list: (kind, meta, start event, end event) of attention / head launches


class _Bracket:
    """HIP events around one launch on the current stream, recorded only while bench.py's replay asks for them."""

    def __init__(self, kind, meta):
        self.kind, self.meta, self.on = kind, meta, KERNEL_PROFILE is not None

    def __enter__(self):
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on and exc[0] is None:
            self.e1.record()
            KERNEL_PROFILE.append((self.kind, self.meta, self.e0, self.e1))
        return False


def _esz(dt):
    return 2 if dt == BF16 else 4


GEMM_SHAPES = None      # tools / bench.py (MMRCA_BENCH_SHAPES=1) set this to a list: (M, N, K, a_layout, b_layout, accum, bytes, e0, e1) of EVERY mmrca_gemm launch
CONV_PROFILE = None     # bench.py sets this to a list: (kernel family, algorithmic HBM bytes, start event, end event) of every conv-path launch


_STREAMK_WS = {}       # (device index, stream) -> zero-filled workspace registered with mmrca_gemm_streamk_workspace
STREAMK = os.environ.get("MMRCA_SK", "1") != "0"     # the library applies it to the fused bf16x3 products only ...
STREAMK_BF16 = os.environ.get("MMRCA_SK_BF16", "0") == "1"   # ... unless this asks for the plain bf16 products too (measured neutral)


def streamk_workspace(M, N, device, force=False):
    """The persistent 256x256 GEMM keeps the partial round of a launch inside the launch (stream-K tail, csrc/gemm256.hip) when the
    stream it runs on has a workspace: one per (device, stream), allocated at the first product that has a full round of tiles.
    Without a workspace there is no tail: MMRCA_SK=0 stops the registration (force=True: tests, tools/streamk_bench.py)."""
    if not (STREAMK or force) or N % 256 or ((M + 255) // 256) * (N // 256) < 256:
        return
    st = stream_ptr()
    key = (device.index if device.index is not None else torch.cuda.current_device(), st)
    if key in _STREAMK_WS:
        return
    if not force and torch.cuda.is_current_stream_capturing():
        return      # a capture stream meets its first large product: no allocation / fill inside the capture -- it keeps the two-launch split
    need = int(load().mmrca_gemm_streamk_workspace_bytes())
    ws = torch.zeros(need, dtype=torch.uint8, device=device)
    _check(load().mmrca_gemm_streamk_workspace(ptr(ws), need, st), "mmrca_gemm_streamk_workspace")
    _STREAMK_WS[key] = ws


def gemm_is_mfma(M, N, K, a_layout, dtype, impl):
    return dtype == BF16 and impl != IMPL_REF and N % 128 == 0 and K % 64 == 0 and (a_layout == ROWK or M % 128 == 0)


def gemm(A, B, Cout, *, bias=None, addend=None, preact=None, M, N, K, lda, ldb, ldc, a_layout=ROWK, b_layout=ROWK,
         act=ACT_NONE, accum=False, dtype, impl=IMPL_AUTO, colsum=None, rows_readable=None):
    """colsum (fp32 [N], +=): column sums of the stored C ride on the GEMM epilogue (mmrca_gemm_colsum).
    rows_readable = (rows of A, rows of the side operand) that exist in memory: required for impl=IMPL_MFMA256 with a ragged M
    (mmrca_gemm_rows; the 256x256 kernel reads whole 256-row tiles)"""
    _dev(A, "gemm A")
    if STREAMK_BF16 and dtype == BF16 and not accum:
        streamk_workspace(M, N, A.device)
    if rows_readable is not None:
        if accum:
            raise MmrcaError("gemm: rows_readable is not available in accumulate mode")
        _check(load().mmrca_gemm_rows(ptr(A), ptr(B), ptr(Cout), ptr(bias), ptr(addend), ptr(preact), ptr(colsum), M, N, K, lda, ldb, ldc,
                                      int(rows_readable[0]), int(rows_readable[1]), a_layout, b_layout, act, dtype, impl, stream_ptr()),
               "mmrca_gemm_rows")
        return
    # matrix-core launches: the bf16 MFMA kernels, and in fp32 mode the general kernel on the fp32 matrix cores
    prof = GEMM_PROFILE is not None and (gemm_is_mfma(M, N, K, a_layout, dtype, impl) or (dtype == F32 and impl != IMPL_REF))
    timed = prof or CONV_PROFILE is not None      # the conv step's byte table times EVERY GEMM (ragged / general-kernel shapes too)
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if colsum is not None:
        if accum:
            raise MmrcaError("gemm: colsum is not available in accumulate mode")
        _check(load().mmrca_gemm_colsum(ptr(A), ptr(B), ptr(Cout), ptr(bias), ptr(addend), ptr(preact), ptr(colsum), M, N, K,
                                        lda, ldb, ldc, a_layout, b_layout, act, dtype, impl, stream_ptr()), "mmrca_gemm_colsum")
    else:
        _check(load().mmrca_gemm(ptr(A), ptr(B), ptr(Cout), ptr(bias), ptr(addend), ptr(preact), M, N, K, lda, ldb, ldc,
                                 a_layout, b_layout, act, int(accum), dtype, impl, stream_ptr()), "mmrca_gemm")
    if timed:
        e1.record()
    if prof:
        GEMM_PROFILE.append((2.0 * M * N * K, (a_layout, b_layout, int(accum)), e0, e1, (M, N, K, act)))
    if CONV_PROFILE is not None:         # byte accounting of the conv step (every GEMM, matrix-core qualified or not)
        es = _esz(dtype)
        side = (1 if addend is not None else 0) + (1 if preact is not None else 0)
        nbytes = int((M * K + N * K) * es + M * N * ((4 if accum else es) + side * es))
        CONV_PROFILE.append(("GEMM (1x1 conv / patch matrix / text encoder)", nbytes, e0, e1))
        if GEMM_SHAPES is not None:
            GEMM_SHAPES.append((M, N, K, a_layout, b_layout, int(accum), nbytes, e0, e1))


SPLITK_WS_BYTES = 64 << 20      # 256 partial tiles of 256x256 fp32: enough for every shape mmrca_gemm_splitk accepts


def gemm_splitk_ok(M, N, K, dtype):
    return dtype == BF16 and M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and K >= 128 and (M // 256) * (N // 256) <= 256


def gemm_splitk_ragged_ok(M, N, K, dtype):
    """the same kernel on output shapes that are not multiples of 256 (K-major bf16 operands only: the conv layers' weight gradients,
    176 x 1056, 48 x 192, ...): edge tiles compute on duplicated columns and the reduction stores only what lies inside C"""
    return (dtype == BF16 and M % 8 == 0 and N % 8 == 0 and K % 64 == 0 and K >= 128
            and ((M + 255) // 256) * ((N + 255) // 256) <= 256)


def gemm_splitk(A, B, Cout, workspace, *, M, N, K, lda, ldb, ldc, a_layout=KROW, b_layout=KROW):
    """fp32 Cout += A (.) B over K on 256x256 tiles, partial tiles through `workspace` (uint8/any dtype tensor in HBM)"""
    _dev(A, "gemm_splitk A")
    prof = GEMM_PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(load().mmrca_gemm_splitk(ptr(A), ptr(B), ptr(Cout), ptr(workspace), workspace.numel() * workspace.element_size(),
                                    M, N, K, lda, ldb, ldc, a_layout, b_layout, stream_ptr()), "mmrca_gemm_splitk")
    if prof:
        e1.record()
        GEMM_PROFILE.append((2.0 * M * N * K, (a_layout, b_layout, 1), e0, e1, (M, N, K, ACT_NONE)))
    if CONV_PROFILE is not None:
        CONV_PROFILE.append(("GEMM (1x1 conv / patch matrix / text encoder)", int((M * K + N * K) * 2 + M * N * 4), None, None))


def gemm_x3(A, B, Cout, *, C_lo=None, bias=None, addend=None, preact=None, colsum=None, M, N, K, lda, ldb, ldc, a_layout=ROWK,
            b_layout=ROWK, act=ACT_NONE, accum=False, impl=IMPL_AUTO):
    """bf16x3 product (csrc/gemm_x3.hip): A and B are (hi, lo) pairs of bf16 planes of fp32 operands; Cout / bias / addend /
    preact are fp32 -- or, with C_lo, the output is written as two bf16 planes (Cout = hi plane)."""
    _dev(A[0], "gemm_x3 A")
    if not accum:
        streamk_workspace(M, N, A[0].device)
    prof = GEMM_PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(load().mmrca_gemm_x3(ptr(A[0]), ptr(A[1]), ptr(B[0]), ptr(B[1]), ptr(Cout), ptr(C_lo), ptr(bias), ptr(addend), ptr(preact),
                                ptr(colsum), M, N, K, lda, ldb, ldc, a_layout, b_layout, act, int(accum), impl, stream_ptr()),
           "mmrca_gemm_x3")
    if prof:
        e1.record()
        # three bf16 MFMA passes per product: the executed matrix-core work is 3 x 2MNK
        GEMM_PROFILE.append((6.0 * M * N * K, (a_layout, b_layout, int(accum)), e0, e1, (M, N, K, act)))


def gemm_splitk_x3(A, B, Cout, workspace, *, M, N, K, lda, ldb, ldc, a_layout=KROW, b_layout=KROW):
    """fp32 Cout += A (.) B over K, bf16x3 operands ((hi, lo) plane pairs), 256x256 tiles with the virtual 3K contraction split
    over the CUs (partial tiles through `workspace`, no atomics)"""
    _dev(A[0], "gemm_splitk_x3 A")
    prof = GEMM_PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(load().mmrca_gemm_splitk_x3(ptr(A[0]), ptr(A[1]), ptr(B[0]), ptr(B[1]), ptr(Cout), ptr(workspace),
                                       workspace.numel() * workspace.element_size(), M, N, K, lda, ldb, ldc, a_layout, b_layout,
                                       stream_ptr()), "mmrca_gemm_splitk_x3")
    if prof:
        e1.record()
        GEMM_PROFILE.append((6.0 * M * N * K, (a_layout, b_layout, 1), e0, e1, (M, N, K, ACT_NONE)))


def split_f32(src, hi, lo, n):
    """hi = bf16(src), lo = bf16(src - hi): the two-plane form of fp32 values the bf16x3 GEMMs read"""
    _dev(src, "split src")
    _check(load().mmrca_split_f32(ptr(src), ptr(hi), ptr(lo), n, stream_ptr()), "mmrca_split_f32")


def colsum_accum(dY, db, M, N, ld, dtype):
    _check(load().mmrca_colsum_accum(ptr(dY), ptr(db), M, N, ld, dtype, stream_ptr()), "mmrca_colsum_accum")


def gelu_bwd(dG, H, dH, n, dtype):
    _check(load().mmrca_gelu_bwd(ptr(dG), ptr(H), ptr(dH), n, dtype, stream_ptr()), "mmrca_gelu_bwd")


def gelu_bwd_colsum(dG, H, dH, db, M, N, ld, dtype):
    _check(load().mmrca_gelu_bwd_colsum(ptr(dG), ptr(H), ptr(dH), ptr(db), M, N, ld, dtype, stream_ptr()), "mmrca_gelu_bwd_colsum")


def mha_fwd(qkv, key_mask, out, lse, B, H, S, dh, scale, dtype, impl=IMPL_AUTO, drop_p=0.0, drop_seed=0, cu=None):
    """cu (int32 [B+1], device): packed token layout, sequence b = rows [cu[b], cu[b+1]); None = padded [B*S] rows"""
    _dev(qkv, "mha qkv")
    with _Bracket("mha_fwd", (B, H, S, dh, cu is not None)):
        _check(load().mmrca_mha_fwd(ptr(qkv), ptr(key_mask), ptr(out), ptr(lse), B, H, S, dh, scale, drop_p, drop_seed, ptr(cu), dtype,
                                    impl, stream_ptr()), "mmrca_mha_fwd")


def mha_fwd_planes_ok(S, dh):
    return dh == 64 and 1 <= S <= 208 and os.environ.get("MMRCA_ATTN_F32_MFMA", "1") != "0"


def mha_fwd_planes(qkv, key_mask, out, out_planes, lse, B, H, S, dh, scale, drop_p=0.0, drop_seed=0, cu=None):
    """fp32 attention forward that also writes the context as (hi, lo) bf16 planes (bf16x3 mode)"""
    _dev(qkv, "mha qkv")
    with _Bracket("mha_fwd", (B, H, S, dh, cu is not None)):
        _check(load().mmrca_mha_fwd_planes(ptr(qkv), ptr(key_mask), ptr(out), ptr(out_planes[0]), ptr(out_planes[1]), ptr(lse), B, H, S, dh,
                                           scale, drop_p, drop_seed, ptr(cu), stream_ptr()), "mmrca_mha_fwd_planes")


def mha_fwd_planes_in(qkv_planes, key_mask, out, out_planes, lse, B, H, S, dh, scale, drop_p=0.0, drop_seed=0, cu=None):
    """mha_fwd_planes with q|k|v given as (hi, lo) bf16 planes (bf16x3f mode)"""
    _dev(qkv_planes[0], "mha qkv")
    with _Bracket("mha_fwd", (B, H, S, dh, cu is not None)):
        _check(load().mmrca_mha_fwd_planes_in(ptr(qkv_planes[0]), ptr(qkv_planes[1]), ptr(key_mask), ptr(out), ptr(out_planes[0]),
                                              ptr(out_planes[1]), ptr(lse), B, H, S, dh, scale, drop_p, drop_seed, ptr(cu), stream_ptr()),
               "mmrca_mha_fwd_planes_in")


def mha_fwd_x3_ok(S, dh):
    return dh == 64 and 1 <= S <= 224 and os.environ.get("MMRCA_ATTN_X3", "1") != "0"


def mha_fwd_x3(qkv_planes, key_mask, out_planes, lse, B, H, S, dh, scale, drop_p=0.0, drop_seed=0, cu=None):
    """bf16x3 attention forward on the bf16 matrix cores: q|k|v and the context as (hi, lo) bf16 planes"""
    _dev(qkv_planes[0], "mha qkv")
    with _Bracket("mha_fwd", (B, H, S, dh, cu is not None)):
        _check(load().mmrca_mha_fwd_x3(ptr(qkv_planes[0]), ptr(qkv_planes[1]), ptr(key_mask), ptr(out_planes[0]), ptr(out_planes[1]), ptr(lse),
                                       B, H, S, dh, scale, drop_p, drop_seed, ptr(cu), stream_ptr()), "mmrca_mha_fwd_x3")


def mha_bwd(qkv, key_mask, out, dout, lse, dqkv, B, H, S, dh, scale, dtype, impl=IMPL_AUTO, drop_p=0.0, drop_seed=0, colsum=None,
            cu=None, rows=None):
    """colsum (fp32 [3*H*dh], +=): column sums of dqkv = the in-projection bias gradient, reduced inside the kernels;
    rows = number of token rows (defaults to B*S; pass cu[B] for the packed layout)"""
    with _Bracket("mha_bwd", (B, H, S, dh, cu is not None)):
        if colsum is not None:
            _check(load().mmrca_mha_bwd_colsum(ptr(qkv), ptr(key_mask), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(colsum),
                                               B * S if rows is None else rows, B, H, S, dh, scale, drop_p, drop_seed, ptr(cu), dtype,
                                               impl, stream_ptr()), "mmrca_mha_bwd_colsum")
            return
        _check(load().mmrca_mha_bwd(ptr(qkv), ptr(key_mask), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), B, H, S, dh, scale,
                                    drop_p, drop_seed, ptr(cu), dtype, impl, stream_ptr()), "mmrca_mha_bwd")


def mha_cross_fwd(q, ldq, k, ldk, v, ldv, out, ldo, B, H, Sq, Skv, dh, scale, dtype, impl=IMPL_AUTO, drop_p=0.0, drop_seed=0):
    """attention forward with separate q / k / v operands (tensors or views; the pointers are taken at their first element)"""
    _dev(q, "q"); _dev(k, "k"); _dev(v, "v"); _dev(out, "out")
    _check(load().mmrca_mha_cross_fwd(ptr(q), ldq, ptr(k), ldk, ptr(v), ldv, ptr(out), ldo, B, H, Sq, Skv, dh, scale, drop_p,
                                      drop_seed, dtype, impl, stream_ptr()), "mha_cross_fwd")


def mha_cross_fwd_x3(q, ldq, k, ldk, v, ldv, out_planes, ldo, B, H, Sq, Skv, dh, scale, drop_p=0.0, drop_seed=0):
    """bf16x3 attention forward with separate fp32 q / k / v operands; the context as (hi, lo) bf16 planes"""
    _dev(q, "q"); _dev(k, "k"); _dev(v, "v"); _dev(out_planes[0], "out")
    if q.dtype != torch.float32 or k.dtype != torch.float32 or v.dtype != torch.float32 or out_planes[0].dtype != torch.bfloat16:
        raise MmrcaError("mha_cross_fwd_x3: q / k / v must be fp32 and the output planes bf16")
    _check(load().mmrca_mha_cross_fwd_x3(ptr(q), ldq, ptr(k), ldk, ptr(v), ldv, ptr(out_planes[0]), ptr(out_planes[1]), ldo, B, H, Sq, Skv, dh,
                                         scale, drop_p, drop_seed, stream_ptr()), "mmrca_mha_cross_fwd_x3")


def mha_cls_fwd(qkv, key_mask, out, lse, B, H, S, dh, scale, dtype, drop_p=0.0, drop_seed=0, cu=None):
    """attention of the class-token query (row 0) only: out [B, H*dh], lse [B, H]"""
    _dev(qkv, "mha_cls qkv")
    _check(load().mmrca_mha_cls_fwd(ptr(qkv), ptr(key_mask), ptr(out), ptr(lse), B, H, S, dh, scale, drop_p, drop_seed, ptr(cu),
                                    dtype, stream_ptr()), "mmrca_mha_cls_fwd")


def mha_cls_bwd(qkv, key_mask, out, dout, lse, dqkv, B, H, S, dh, scale, dtype, drop_p=0.0, drop_seed=0, cu=None):
    _check(load().mmrca_mha_cls_bwd(ptr(qkv), ptr(key_mask), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), B, H, S, dh, scale,
                                    drop_p, drop_seed, ptr(cu), dtype, stream_ptr()), "mmrca_mha_cls_bwd")


def add_layernorm_fwd(x, res, gamma, beta, sum_out, y, mean, rstd, rows, D, ld_x, ld_y, eps, dtype,
                      in_drop=(0.0, 0), out_drop=(0.0, 0), y_planes=None):
    """y_planes = (hi, lo) bf16 tensors (fp32 operands only): the output also -- or, with y None, only -- as two bf16 planes"""
    _dev(x, "layernorm x")
    if y_planes is not None:
        _check(load().mmrca_add_layernorm_fwd_x3(ptr(x), ptr(res), ptr(gamma), ptr(beta), ptr(sum_out), ptr(y), ptr(y_planes[0]),
                                                 ptr(y_planes[1]), ptr(mean), ptr(rstd), rows, D, ld_x, ld_y, eps, in_drop[0], in_drop[1],
                                                 out_drop[0], out_drop[1], stream_ptr()), "mmrca_add_layernorm_fwd_x3")
        return
    _check(load().mmrca_add_layernorm_fwd(ptr(x), ptr(res), ptr(gamma), ptr(beta), ptr(sum_out), ptr(y), ptr(mean), ptr(rstd),
                                          rows, D, ld_x, ld_y, eps, in_drop[0], in_drop[1], out_drop[0], out_drop[1],
                                          dtype, stream_ptr()), "mmrca_add_layernorm_fwd")


def layernorm_bwd(dy, s, gamma, mean, rstd, dres, ds, dgamma, dbeta, rows, D, ld_dy, ld_s, ld_ds, dtype,
                  dy_drop=(0.0, 0), branch_drop=(0.0, 0), dbranch=None, dcol=None, dcol_branch=None):
    _check(load().mmrca_layernorm_bwd(ptr(dy), ptr(s), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(ds), ptr(dgamma),
                                      ptr(dbeta), rows, D, ld_dy, ld_s, ld_ds, dy_drop[0], dy_drop[1], branch_drop[0],
                                      branch_drop[1], ptr(dbranch), ptr(dcol), ptr(dcol_branch), dtype, stream_ptr()), "mmrca_layernorm_bwd")


def layernorm_bwd_mixed(dy, s, gamma, mean, rstd, dres, ds, dgamma, dbeta, rows, D, ld_dy, ld_s, ld_ds,
                        dy_drop=(0.0, 0), branch_drop=(0.0, 0), dbranch=None, dcol=None, dcol_branch=None):
    """layernorm_bwd with bf16 gradients / gamma against the fp32 saved sum `s` (bf16x3f mode)"""
    _check(load().mmrca_layernorm_bwd_mixed(ptr(dy), ptr(s), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(ds), ptr(dgamma),
                                            ptr(dbeta), rows, D, ld_dy, ld_s, ld_ds, dy_drop[0], dy_drop[1], branch_drop[0],
                                            branch_drop[1], ptr(dbranch), ptr(dcol), ptr(dcol_branch), stream_ptr()), "mmrca_layernorm_bwd_mixed")


def embed_fwd(ids, pos_ids, word, pos, type_row, out, rows, D, dtype):
    _dev(ids, "embed ids")
    _check(load().mmrca_embed_fwd(ptr(ids), ptr(pos_ids), ptr(word), ptr(pos), ptr(type_row), ptr(out), rows, D, dtype, stream_ptr()), "mmrca_embed_fwd")


def embed_bwd(dout, ids, pos_ids, dword, dpos, dtype_row, rows, D, dtype, pad_id=-1, pos_pad_id=-1):
    _check(load().mmrca_embed_bwd(ptr(dout), ptr(ids), ptr(pos_ids), ptr(dword), ptr(dpos), ptr(dtype_row), rows, D,
                                  pad_id, pos_pad_id, dtype, stream_ptr()), "mmrca_embed_bwd")


def patchify_fwd(images, patches, B, Cc, Himg, Wimg, P, dtype):
    _dev(images, "patchify images")
    _check(load().mmrca_patchify_fwd(ptr(images), ptr(patches), B, Cc, Himg, Wimg, P, dtype, stream_ptr()), "mmrca_patchify_fwd")


def vit_assemble_fwd(proj, cls, pos, x, B, nP, D, dtype):
    _check(load().mmrca_vit_assemble_fwd(ptr(proj), ptr(cls), ptr(pos), ptr(x), B, nP, D, dtype, stream_ptr()), "mmrca_vit_assemble_fwd")


def vit_assemble_bwd(dx, dproj, dcls, dpos, B, nP, D, dtype):
    _check(load().mmrca_vit_assemble_bwd(ptr(dx), ptr(dproj), ptr(dcls), ptr(dpos), B, nP, D, dtype, stream_ptr()), "mmrca_vit_assemble_bwd")


def head_fwd(img, txt, w: HeadPtrs, logits, B, d_img, d_txt, n_classes, reverse, mode, drop_p, seed, dtype):
    _dev(img, "head img")
    with _Bracket("head_fwd", (B, d_img, d_txt)):
        _check(load().mmrca_head_fwd(ptr(img), ptr(txt), C.byref(w), ptr(logits), B, d_img, d_txt, n_classes, int(reverse), mode,
                                     drop_p, seed, dtype, stream_ptr()), "mmrca_head_fwd")


_HEAD_WS = {}      # (device index, stream) -> workspace tensor of the head backward (grows on demand)


def head_bwd_workspace_bytes(B, d_img, d_txt) -> int:
    fn = load().mmrca_head_bwd_workspace_bytes
    fn.restype, fn.argtypes = C.c_int64, [_i32, _i32, _i32]
    return int(fn(B, d_img, d_txt))


def head_bwd(dlogits, img, txt, w: HeadPtrs, g: HeadPtrs, dimg, dtxt, B, d_img, d_txt, n_classes, reverse, mode, drop_p, seed, dtype,
             workspace=None):
    """Gradients are accumulated into ``g``.  ``workspace``: uint8/any tensor of ``head_bwd_workspace_bytes`` bytes; by default one
    is kept per (device, stream) -- the two launches of the backward and its next use are ordered by that stream."""
    need = head_bwd_workspace_bytes(B, d_img, d_txt)
    if workspace is None:
        key = (img.device.index, stream_ptr())
        workspace = _HEAD_WS.get(key)
        if workspace is None or workspace.numel() < need:
            workspace = _HEAD_WS[key] = torch.empty(need, dtype=torch.uint8, device=img.device)
    with _Bracket("head_bwd", (B, d_img, d_txt)):
        _check(load().mmrca_head_bwd(ptr(dlogits), ptr(img), ptr(txt), C.byref(w), C.byref(g), ptr(dimg), ptr(dtxt), B, d_img, d_txt,
                                     n_classes, int(reverse), mode, drop_p, seed, dtype, ptr(workspace),
                                     workspace.numel() * workspace.element_size(), stream_ptr()), "mmrca_head_bwd")


def xent_fwd_bwd(logits, labels, class_w, smoothing, loss, dlogits, B, Cc, grad_scale=1.0):
    _dev(logits, "xent logits")
    _check(load().mmrca_xent_fwd_bwd(ptr(logits), ptr(labels), ptr(class_w), smoothing, ptr(loss), ptr(dlogits), B, Cc, grad_scale, stream_ptr()), "mmrca_xent_fwd_bwd")


def sgd_step(p, g, lp, n, lr, wd, grad_scale=1.0, lp_lo=None):
    """lp / lp_lo: bf16 working copy (hi plane) and, in bf16x3 mode, the lo plane, rewritten in the same pass"""
    _dev(p, "sgd params")
    if lp_lo is not None:
        _check(load().mmrca_sgd_step_x3(ptr(p), ptr(g), ptr(lp), ptr(lp_lo), n, lr, wd, grad_scale, stream_ptr()), "mmrca_sgd_step_x3")
        return
    _check(load().mmrca_sgd_step(ptr(p), ptr(g), ptr(lp), n, lr, wd, grad_scale, stream_ptr()), "mmrca_sgd_step")


def adamw_step(p, g, m, v, lp, n, lr, b1, b2, eps, wd, step, grad_scale=1.0, lp_lo=None):
    _dev(p, "adamw params")
    if lp_lo is not None:
        _check(load().mmrca_adamw_step_x3(ptr(p), ptr(g), ptr(m), ptr(v), ptr(lp), ptr(lp_lo), n, lr, b1, b2, eps, wd, step, grad_scale,
                                          stream_ptr()), "mmrca_adamw_step_x3")
        return
    _check(load().mmrca_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(lp), n, lr, b1, b2, eps, wd, step, grad_scale, stream_ptr()), "mmrca_adamw_step")


SEED_EPOCH_STRIDE = 1000003      # = MMRCA_SEED_EPOCH_STRIDE (csrc/common.h) = the per-step stride of engine._site_seed
_seed_epoch = 0                  # host view of the device-side mask epoch (0 = eager launches draw the seeds they are given)


def seed_epoch_set(value: int = 0, device_value: Optional[torch.Tensor] = None):
    """Mask epoch of every dropout kernel (include/mmrca.h).  device_value: int64 [1] tensor in HBM, read when the launch executes
    (captured in a HIP graph it follows the tensor from replay to replay); else the immediate `value`."""
    global _seed_epoch
    if device_value is not None:
        _dev(device_value, "seed_epoch_set")
        if device_value.dtype != torch.int64 or device_value.numel() < 1:
            raise MmrcaError("seed_epoch_set: device_value must be an int64 tensor")
    _check(load().mmrca_seed_epoch_set(ptr(device_value), int(value) & 0xFFFFFFFFFFFFFFFF, stream_ptr()), "mmrca_seed_epoch_set")
    _seed_epoch = None if device_value is not None else int(value)


def sd_rowscale(p: torch.Tensor, out: torch.Tensor, n: int, B: int, seed: int):
    """stochastic-depth row scales of n residual blocks x B samples (include/mmrca.h): out = keep / (1 - p), fp32"""
    _dev(p, "sd_rowscale"); _dev(out, "sd_rowscale")
    if p.dtype != torch.float32 or out.dtype != torch.float32 or p.numel() < n or out.numel() < n * B:
        raise MmrcaError("sd_rowscale: p [n] and out [n, B] must be fp32 tensors of at least those sizes")
    _check(load().mmrca_sd_rowscale(ptr(p), ptr(out), int(n), int(B), int(seed) & 0xFFFFFFFFFFFFFFFF, stream_ptr()), "mmrca_sd_rowscale")


def seed_epoch_host():
    """the epoch the device holds as far as the host knows (None: it follows a device counter -- GraphedTrainStep keeps track)"""
    return _seed_epoch


def seed_epoch_note(value):
    global _seed_epoch
    _seed_epoch = value


def cast_f32_to_bf16(src, dst, n):
    _dev(src, "cast src")
    _check(load().mmrca_cast_f32_to_bf16(ptr(src), ptr(dst), n, stream_ptr()), "mmrca_cast_f32_to_bf16")


# ---------------------------------------------------------------------------------------------------------
# conv-backbone kernels (csrc/conv.hip); thin wrappers, raw pointers in
# ---------------------------------------------------------------------------------------------------------
CONV_NONE, CONV_SILU, CONV_RELU, CONV_SIGMOID = 0, 1, 2, 3


def _out_hw(H, W, s):
    return (H - 1) // s + 1, (W - 1) // s + 1


def _patch_bytes(a):          # (x, col, B, H, W, C, stride, ldk, dtype): the input rows once + the patch matrix once
    Ho, Wo = _out_hw(a[3], a[4], a[6])
    return (a[2] * a[3] * a[4] * a[5] + a[2] * Ho * Wo * a[7]) * _esz(a[8])


def _dw_bwd_bytes(a):         # (dy, x, w, dx, dw, B, H, W, C, stride, dtype, ...): dy (output rows) + x + dx (input rows)
    Ho, Wo = _out_hw(a[6], a[7], a[9])
    return a[5] * a[8] * (Ho * Wo + 2 * a[6] * a[7]) * _esz(a[10])


# ALGORITHMIC HBM bytes of one launch (every operand the op must read once + every result it must write once, per-channel
# vectors ignored), from the launch arguments: the byte roofline of the conv step in bench.py
_CONV_BYTES = {
    "mmrca_nchw_to_rows": ("layout", lambda a: a[2] * a[3] * a[4] * a[5] * (4 + _esz(a[6]))),
    "mmrca_im2row3x3": ("patches", _patch_bytes), "mmrca_col2im3x3": ("patches", _patch_bytes),
    "mmrca_im2row3x3_tap": ("patches", _patch_bytes), "mmrca_col2im3x3_tap": ("patches", _patch_bytes),
    "mmrca_dwconv3x3_fwd": ("depthwise", lambda a: a[3] * a[6] * (a[4] * a[5] + _out_hw(a[4], a[5], a[7])[0] * _out_hw(a[4], a[5], a[7])[1]) * _esz(a[8])),
    "mmrca_dwconv3x3_bwd": ("depthwise", _dw_bwd_bytes), "mmrca_dwconv3x3_bwd_ws": ("depthwise", _dw_bwd_bytes),
    "mmrca_conv3x3_fwd": ("conv3x3 implicit GEMM", lambda a: (a[6] * a[7] * a[8] * (a[9] + a[10]) + 9 * a[9] * a[10]) * _esz(a[11])),
    "mmrca_conv3x3_wgrad": ("conv3x3 implicit GEMM", lambda a: a[3] * a[4] * a[5] * (a[6] + a[7]) * _esz(a[8]) + 9 * a[6] * a[7] * 4),
    "mmrca_gemm_bnstats": ("GEMM (1x1 / patch)", lambda a: (a[3] * a[5] + a[4] * a[5] + a[3] * a[4]) * _esz(a[9])),
    "mmrca_channel_interleave2": ("layout", lambda a: 2 * a[4] * 2 * a[5] * _esz(a[6])),
    "mmrca_channel_deinterleave2": ("layout", lambda a: 2 * a[4] * 2 * a[5] * _esz(a[6])),
    "mmrca_bn_stats": ("BatchNorm", lambda a: a[5] * a[6] * _esz(a[11]) if a[10] else 0),
    "mmrca_bn_stats_ws": ("BatchNorm", lambda a: a[5] * a[6] * _esz(a[11]) if a[10] else 0),
    "mmrca_bn_stats_fused": ("BatchNorm", lambda a: a[5] * a[6] * _esz(a[10])),
    "mmrca_bn_act_fwd": ("BatchNorm", lambda a: 2 * a[6] * a[7] * _esz(a[9])),
    "mmrca_bn_act_fwd_res": ("BatchNorm", lambda a: 3 * a[8] * a[9] * _esz(a[12])),
    "mmrca_bn_moments": ("BatchNorm", lambda a: a[4] * a[5] * _esz(a[7])),
    "mmrca_bn_act_fwd_fin": ("BatchNorm", lambda a: (3 if a[6] else 2) * a[13] * a[14] * _esz(a[19])),
    # backward with batch statistics: the sums need dy and z once, the apply pass needs them again and writes dx
    "mmrca_bn_act_bwd": ("BatchNorm", lambda a: (5 if a[13] else 3) * a[10] * a[11] * _esz(a[14])),
    "mmrca_bn_act_bwd_ws": ("BatchNorm", lambda a: (5 if a[13] else 3) * a[10] * a[11] * _esz(a[14])),
    "mmrca_bn_act_bwd_sums": ("BatchNorm", lambda a: 3 * a[10] * a[11] * _esz(a[14])),
    "mmrca_rowpool_mean": ("pool / squeeze-excitation", lambda a: a[2] * a[3] * a[4] * _esz(a[5])),
    "mmrca_rowpool_mean_bwd": ("pool / squeeze-excitation", lambda a: a[2] * a[3] * a[4] * _esz(a[6]) * (2 if a[5] else 1)),
    "mmrca_se_scale_fwd": ("pool / squeeze-excitation", lambda a: 2 * a[3] * a[4] * a[5] * _esz(a[6])),
    "mmrca_se_mlp_fwd": ("pool / squeeze-excitation", lambda a: (2 * a[10] * a[11] + 3 * a[9] * (a[10] + a[11])) * _esz(a[12])),
    "mmrca_se_mlp_bwd": ("pool / squeeze-excitation", lambda a: (2 * a[17] * a[18] + 6 * a[16] * (a[17] + a[18])) * _esz(a[19]) + 2 * a[17] * a[18] * 4),
    "mmrca_se_scale_bwd": ("pool / squeeze-excitation", lambda a: (3 if a[3] else 2) * a[5] * a[6] * a[7] * _esz(a[8])),
    "mmrca_se_dx": ("pool / squeeze-excitation", lambda a: (3 if a[8] else 2) * a[4] * a[5] * a[6] * _esz(a[7])),
    "mmrca_residual_add": ("residual", lambda a: (3 if a[0] else 2) * a[4] * a[5] * _esz(a[6])),
    "mmrca_maxpool3x3s2_fwd": ("pool / squeeze-excitation", lambda a: a[3] * a[6] * (a[4] * a[5] * _esz(a[7]) + _out_hw(a[4], a[5], 2)[0] * _out_hw(a[4], a[5], 2)[1] * (_esz(a[7]) + 1))),
    "mmrca_maxpool3x3s2_bwd": ("pool / squeeze-excitation", lambda a: a[3] * a[6] * (a[4] * a[5] * _esz(a[7]) + _out_hw(a[4], a[5], 2)[0] * _out_hw(a[4], a[5], 2)[1] * (_esz(a[7]) + 1))),
    "mmrca_channel_gather": ("layout", lambda a: 2 * a[3] * a[5] * _esz(a[8])),
}


def _c(name, *args):
    if CONV_PROFILE is not None:
        fam, fn = _CONV_BYTES.get(name, ("other", lambda a: 0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _check(getattr(load(), name)(*args, stream_ptr()), name)
        e1.record()
        CONV_PROFILE.append((fam, int(fn(args)), e0, e1))
        return
    _check(getattr(load(), name)(*args, stream_ptr()), name)


def nchw_to_rows(images, x, B, C, H, W, dtype):
    _dev(images, "nchw_to_rows images")
    _c("mmrca_nchw_to_rows", ptr(images), ptr(x), B, C, H, W, dtype)


def im2row3x3(x, col, B, H, W, C, stride, ldk, dtype):
    _dev(x, "im2row x")
    _c("mmrca_im2row3x3", ptr(x), ptr(col), B, H, W, C, stride, ldk, dtype)


def col2im3x3(dcol, dx, B, H, W, C, stride, ldk, dtype):
    _c("mmrca_col2im3x3", ptr(dcol), ptr(dx), B, H, W, C, stride, ldk, dtype)


def im2row3x3_tap(x, col, B, H, W, C, stride, ldk, dtype):
    """tap-major patches (k = tap*C + c); bf16, C % 8 == 0"""
    _dev(x, "im2row x")
    _c("mmrca_im2row3x3_tap", ptr(x), ptr(col), B, H, W, C, stride, ldk, dtype)


def col2im3x3_tap(dcol, dx, B, H, W, C, stride, ldk, dtype):
    _c("mmrca_col2im3x3_tap", ptr(dcol), ptr(dx), B, H, W, C, stride, ldk, dtype)


def dwconv3x3_fwd(x, w, y, B, H, W, C, stride, dtype):
    _dev(x, "dwconv x")
    _c("mmrca_dwconv3x3_fwd", ptr(x), ptr(w), ptr(y), B, H, W, C, stride, dtype)


def dwconv3x3_bwd(dy, x, w, dx, dw, B, H, W, C, stride, dtype, ws=None):
    """ws: optional scratch tensor for the weight gradient's per-block partial sums (replaces its fp32 atomics)"""
    if ws is None:
        _c("mmrca_dwconv3x3_bwd", ptr(dy), ptr(x), ptr(w), ptr(dx), ptr(dw), B, H, W, C, stride, dtype)
    else:
        _c("mmrca_dwconv3x3_bwd_ws", ptr(dy), ptr(x), ptr(w), ptr(dx), ptr(dw), B, H, W, C, stride, dtype, ptr(ws), ws.numel() * ws.element_size())


def conv3x3_stat_slots(B, H, W) -> int:
    """rows of the (mean, M2) slot arrays / length of the count array conv3x3_fwd fills for a [B, H, W] output"""
    return int(load().mmrca_conv3x3_stat_slots(B, H, W))


def conv3x3_fwd(x, w_tap, z, B, H, W, Cin, Cout, dtype, parts=None):
    """implicit-GEMM 3x3 / stride 1 / pad 1 convolution (no patch matrix); parts = (part_mean, part_m2, part_cnt) collects the
    BatchNorm moments of z in the epilogue (conv_bn_finish turns them into mean / rstd)"""
    pm, p2, pc = parts if parts is not None else (None, None, None)
    _c("mmrca_conv3x3_fwd", ptr(x), ptr(w_tap), ptr(z), ptr(pm), ptr(p2), ptr(pc), B, H, W, Cin, Cout, dtype)


def conv_bn_finish(parts, B, H, W, mean, rstd, running_mean, running_var, C, eps, momentum):
    _c("mmrca_conv_bn_finish", ptr(parts[0]), ptr(parts[1]), ptr(parts[2]), B, H, W, ptr(mean), ptr(rstd), ptr(running_mean),
       ptr(running_var), C, eps, momentum)


def conv3x3_wgrad(dz, x, dw_tap, B, H, W, Cin, Cout, dtype):
    """dw_tap[Cout, 9*Cin] (fp32) += the weight gradient of the same convolution, patches gathered from x inside the kernel"""
    _c("mmrca_conv3x3_wgrad", ptr(dz), ptr(x), ptr(dw_tap), B, H, W, Cin, Cout, dtype)


def gemm_bnstats_ok(M, N, K, dtype):
    return dtype == BF16 and N % 8 == 0 and K % 32 == 0 and M >= 64


def gemm_bnstats(A, B, Cout, *, M, N, K, lda, ldb, ldc, dtype, shift, s1, s2):
    """plain bf16 GEMM with the BatchNorm moments of its output in the epilogue (s1 / s2: [ceil(M / 128), N] fp32, written)"""
    _c("mmrca_gemm_bnstats", ptr(A), ptr(B), ptr(Cout), M, N, K, lda, ldb, ldc, dtype, ptr(shift), ptr(s1), ptr(s2))


def bn_finish_sums(s1, s2, shift, nslots, rows, mean, rstd, running_mean, running_var, C, eps, momentum):
    _c("mmrca_bn_finish_sums", ptr(s1), ptr(s2), ptr(shift), nslots, rows, ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), C, eps,
       momentum)


def bn_fold_ok(C, ld, dtype):
    """can the train-mode BatchNorm forward run as mmrca_bn_moments + mmrca_bn_act_fwd_fin (two launches instead of three)?"""
    return dtype == BF16 and C % 8 == 0 and ld % 8 == 0


def bn_moments(x, s1, s2, shift, rows, C, ld, dtype):
    """shifted one-pass sums of x[rows, C] into s1 / s2 (+=, caller-zeroed) and the shift into `shift`"""
    _c("mmrca_bn_moments", ptr(x), ptr(s1), ptr(s2), ptr(shift), rows, C, ld, dtype)


def bn_act_fwd_fin(x, s1, s2, shift, gamma, beta, res, rowscale, y, mean, rstd, running_mean, running_var, rows, C, act, rows_per_sample,
                   eps, momentum, dtype):
    """y = [res + rowscale *] act(bn(x)) with the finish step inside: mean / rstd from (s1, s2, shift) -> mean / rstd, running statistics"""
    _c("mmrca_bn_act_fwd_fin", ptr(x), ptr(s1), ptr(s2), ptr(shift), ptr(gamma), ptr(beta), ptr(res), ptr(rowscale), ptr(y), ptr(mean), ptr(rstd),
       ptr(running_mean), ptr(running_var), rows, C, act, rows_per_sample, eps, momentum, dtype)


def bn_act_fwd_res(x, mean, rstd, gamma, beta, res, rowscale, out, rows, C, act, rows_per_sample, dtype):
    """out = res + rowscale[sample] * act(bn(x)): the block's last BatchNorm + activation and its residual connection in one pass"""
    _c("mmrca_bn_act_fwd_res", ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(res), ptr(rowscale), ptr(out), rows, C, act,
       rows_per_sample, dtype)


def bn_stats(x, mean, rstd, running_mean, running_var, rows, C, ld, eps, momentum, train, dtype, ws=None, prezeroed=False, tickets=None):
    """ws: fp32 workspace tensor (the flat streaming reduction of large bf16 tensors, include/mmrca.h); prezeroed: the caller cleared
    mean / rstd (conv_engine's per-step arena): no fill launch; tickets: int32 [ceil(C / 64)] (cleared with them when prezeroed): the
    finish step runs inside the reduction launch (train mode)"""
    if tickets is not None and train and ws is None:
        if tickets.dtype != torch.int32 or tickets.numel() < (C + 63) // 64:
            raise MmrcaError("bn_stats: tickets must be int32 [ceil(C / 64)]")
        _c("mmrca_bn_stats_fused", ptr(x), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), rows, C, ld, eps, momentum, dtype,
           ptr(tickets), int(prezeroed))
        return
    if ws is not None or prezeroed:
        _c("mmrca_bn_stats_ws", ptr(x), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), rows, C, ld, eps, momentum, int(train), dtype,
           ptr(ws), 0 if ws is None else ws.numel() * ws.element_size(), int(prezeroed))
        return
    _c("mmrca_bn_stats", ptr(x), ptr(mean), ptr(rstd), ptr(running_mean), ptr(running_var), rows, C, ld, eps, momentum, int(train), dtype)


def bn_act_fwd(x, mean, rstd, gamma, beta, y, rows, C, act, dtype):
    _dev(x, "bn x")
    _c("mmrca_bn_act_fwd", ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(y), rows, C, act, dtype)


def bn_act_bwd(dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, scratch, rows, C, act, train, dtype, sums_ready=False, ws=None, prezeroed=False):
    """sums_ready: scratch already holds the first pass's sums (se_dx accumulated them): no reduce pass.  ws / prezeroed: as in bn_stats"""
    if (ws is not None or prezeroed) and not sums_ready:
        _c("mmrca_bn_act_bwd_ws", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(scratch),
           rows, C, act, int(train), dtype, ptr(ws), 0 if ws is None else ws.numel() * ws.element_size(), int(prezeroed))
        return
    _c("mmrca_bn_act_bwd_sums" if sums_ready else "mmrca_bn_act_bwd", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(scratch),
       rows, C, act, int(train), dtype)


def rowpool_mean(x, out, B, HW, C, dtype):
    _dev(x, "rowpool x")
    _c("mmrca_rowpool_mean", ptr(x), ptr(out), B, HW, C, dtype)


def rowpool_mean_bwd(dpool, dx, B, HW, C, accumulate, dtype):
    _c("mmrca_rowpool_mean_bwd", ptr(dpool), ptr(dx), B, HW, C, int(accumulate), dtype)


def se_mlp_fwd(pooled, w1, b1, w2, b2, h_pre, h, s_pre, s, B, c, sq, dtype):
    """the squeeze-excitation MLP in one launch (include/mmrca.h)"""
    _c("mmrca_se_mlp_fwd", ptr(pooled), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(h_pre), ptr(h), ptr(s_pre), ptr(s), B, c, sq, dtype)


def se_mlp_bwd(ds, pooled, h_pre, h, s_pre, w1, b1, w2, b2, ds_pre, dh_pre, dpool, gw1, gb1, gw2, gb2, B, c, sq, dtype):
    """... and its backward: dpool, and += into the fp32 parameter gradients"""
    for t in (gw1, gb1, gw2, gb2):
        if t.dtype != torch.float32:
            raise MmrcaError("se_mlp_bwd: parameter gradients must be fp32")
    _c("mmrca_se_mlp_bwd", ptr(ds), ptr(pooled), ptr(h_pre), ptr(h), ptr(s_pre), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(ds_pre), ptr(dh_pre),
       ptr(dpool), ptr(gw1), ptr(gb1), ptr(gw2), ptr(gb2), B, c, sq, dtype)


def se_scale_fwd(x, s, y, B, HW, C, dtype):
    _c("mmrca_se_scale_fwd", ptr(x), ptr(s), ptr(y), B, HW, C, dtype)


def se_scale_bwd(dy, x, s, dx, ds, B, HW, C, dtype):
    """dx may be None (bf16 / C % 8 == 0 only): only ds; se_dx then writes dx with the pooled gradient added"""
    _c("mmrca_se_scale_bwd", ptr(dy), ptr(x), ptr(s), ptr(dx), ptr(ds), B, HW, C, dtype)


def se_dx(dy, s, dpool, dx, B, HW, C, dtype, bn=None):
    """dx = dy * s + dpool / HW in one pass; bn = (z, mean, rstd, gamma, beta, act, sums): also accumulates the BatchNorm-backward
    sums of the layer whose output gradient dx is (bn_act_bwd(..., sums_ready=True) then skips its reduce pass)"""
    z, mean, rstd, gamma, beta, act, sums = bn if bn is not None else (None, None, None, None, None, 0, None)
    _c("mmrca_se_dx", ptr(dy), ptr(s), ptr(dpool), ptr(dx), B, HW, C, dtype, ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, ptr(sums))


def bias_act_fwd(x, bias, y, rows, C, act, dtype):
    _c("mmrca_bias_act_fwd", ptr(x), ptr(bias), ptr(y), rows, C, act, dtype)


def bias_act_bwd(dy, x, bias, dx, dbias, rows, C, act, dtype):
    _c("mmrca_bias_act_bwd", ptr(dy), ptr(x), ptr(bias), ptr(dx), ptr(dbias), rows, C, act, dtype)


def residual_add(a, branch, rowscale, out, B, per_sample, dtype):
    _c("mmrca_residual_add", ptr(a), ptr(branch), ptr(rowscale), ptr(out), B, per_sample, dtype)


def maxpool3x3s2_fwd(x, y, argmax, B, H, W, C, dtype):
    _c("mmrca_maxpool3x3s2_fwd", ptr(x), ptr(y), ptr(argmax), B, H, W, C, dtype)


def maxpool3x3s2_bwd(dy, argmax, dx, B, H, W, C, dtype):
    _c("mmrca_maxpool3x3s2_bwd", ptr(dy), ptr(argmax), ptr(dx), B, H, W, C, dtype)


def channel_interleave2(a, lda, b, out, rows, bf, dtype):
    """out[r, 2j] = a[r, j], out[r, 2j + 1] = b[r, j]: ShuffleNetV2's concat + channel shuffle"""
    _c("mmrca_channel_interleave2", ptr(a), lda, ptr(b), ptr(out), rows, bf, dtype)


def channel_deinterleave2(dout, d_even, ld_even, d_odd, rows, bf, dtype):
    _c("mmrca_channel_deinterleave2", ptr(dout), ptr(d_even), ld_even, ptr(d_odd), rows, bf, dtype)


def channel_gather(inp, cmap, out, rows, Cin, Cout, ld_out, col0, dtype):
    _c("mmrca_channel_gather", ptr(inp), ptr(cmap), ptr(out), rows, Cin, Cout, ld_out, col0, dtype)
