"""ctypes loader for libasq_hip.so (the C-ABI declared in include/asq_hip.h)."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libasq_hip.so")

ASQ_F32, ASQ_F16, ASQ_BF16 = 0, 1, 2
ASQ_ACT_ROUND, ASQ_ACT_DIV, ASQ_ACT_PER_TOKEN = 0, 1, 2
ASQ_EPI_SCALE_FIRST, ASQ_EPI_ACC_FIRST = 0, 1
ASQ_SILU_FAST = 2   # bit 1 of asq_silu_mul_quantize's per_token / asq_linear_w8a8_gate_up's flags
ASQ_FP8_PER_TOKEN, ASQ_FP8_PER_TENSOR, ASQ_FP8_STATIC = 0, 1, 2

ASQ_VERSION = 126   # include/asq_hip.h: the C-ABI this loader was written against
_lock = threading.Lock()
_lib = None

_vp, _i64, _f32, _int, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_size_t

# name -> (restype, argtypes): must list every symbol include/asq_hip.h declares
SIGNATURES = {
    "asq_version": (_int, []),
    "asq_last_error": (ctypes.c_char_p, []),
    "asq_gemm_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "asq_workspace_init": (_int, [_vp, _sz, _vp]),
    "asq_workspace_header_bytes": (_sz, []),
    "asq_gemm_i8_i32": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _sz, _vp]),
    "asq_gemm_i8_i8": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _f32, _f32, _vp, _sz, _vp]),
    "asq_quantize_act": (_int, [_vp, _int, _int, _f32, _vp, _vp, _i64, _i64, _vp]),
    "asq_norm_quantize": (_int, [_vp, _int, _vp, _vp, _f32, _int, _vp, _vp, _i64, _i64, _vp]),
    "asq_add_norm_quantize": (_int, [_vp, _vp, _vp, _int, _vp, _vp, _f32, _int, _vp, _vp, _i64, _i64, _vp]),
    "asq_silu_mul_quantize": (_int, [_vp, _vp, _int, _int, _f32, _vp, _vp, _i64, _i64, _vp]),
    "asq_silu_mul_quantize_fp8": (_int, [_vp, _vp, _int, _int, _vp, _vp, _i64, _i64, _vp]),
    "asq_rope": (_int, [_vp, _i64, _vp, _int, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    "asq_rmsnorm": (_int, [_vp, _int, _vp, _f32, _vp, _i64, _i64, _vp]),
    "asq_silu_mul": (_int, [_vp, _vp, _int, _int, _vp, _i64, _vp]),
    "asq_fp8_grouped_gate_up_supported": (_int, [_i64, _i64, _i64, _int]),
    "asq_linear_fp8_grouped_gate_up": (_int, [_vp, _vp, _vp, _int, _vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp]),
    "asq_linear_w8a8": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _int, _vp, _sz, _vp]),
    "asq_linear_w8a8_q8": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _int, _int, _int, _f32, _vp, _sz, _vp]),
    "asq_linear_w8a8_grouped": (_int, [_vp, _vp, _vp, _int, _vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "asq_grouped_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "asq_linear_w8a8_grouped_ws": (_int, [_vp, _vp, _vp, _int, _vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "asq_linear_w8a8_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "asq_linear_w8a8_forward": (_int, [_vp, _int, _vp, _vp, _i64, _i64, _i64, _int, _f32, _f32, _vp, _vp, _vp, _sz, _vp]),
    "asq_gemm_kernel_name": (ctypes.c_char_p, [_i64, _i64, _i64]),
    "asq_weight_offset_image": (_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "asq_quantize_act_off": (_int, [_vp, _int, _int, _f32, _vp, _vp, _vp, _i64, _i64, _vp]),
    "asq_linear_w8a8_off": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    "asq_offsets_supported": (_int, [_i64, _i64, _i64, _int]),
    "asq_forward_fused_supported": (_int, [_i64, _i64, _i64, _int, _int]),
    "asq_gate_up_supported": (_int, [_i64, _i64, _i64, _int]),
    "asq_linear_w8a8_gate_up": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _f32, _f32, _vp, _int, _vp, _vp, _vp]),
    "asq_linear_w8a8_gate_up_q8": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _f32, _f32, _vp, _int, _f32, _vp, _vp, _vp]),
    "asq_grouped_gate_up_supported": (_int, [_i64, _i64, _i64, _int]),
    "asq_linear_w8a8_grouped_gate_up": (_int, [_vp, _vp, _vp, _int, _vp, _int, _i64, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp, _sz, _vp]),
    "asq_linear_w8a8_forward_fused": (_int, [_vp, _int, _vp, _vp, _i64, _i64, _i64, _int, _f32, _f32, _vp, _vp, _vp]),
    "asq_linear_w8a8_grouped_off": (_int, [_vp, _vp, _vp, _int, _vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "asq_norm_quantize_off": (_int, [_vp, _int, _vp, _vp, _f32, _int, _vp, _vp, _vp, _i64, _i64, _vp]),
    "asq_add_norm_quantize_off": (_int, [_vp, _vp, _vp, _int, _vp, _vp, _f32, _int, _vp, _vp, _vp, _i64, _i64, _vp]),
    "asq_silu_mul_quantize_off": (_int, [_vp, _vp, _int, _int, _f32, _vp, _vp, _vp, _i64, _i64, _vp]),
    "asq_linear_w8a8_forward_off": (_int, [_vp, _int, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _f32, _f32, _vp, _vp, _vp, _sz, _vp]),
    "asq_quantize_act_fp8": (_int, [_vp, _int, _int, _f32, _vp, _vp, _i64, _i64, _vp]),
    "asq_linear_fp8": (_int, [_vp, _vp, _int, _vp, _int, _i64, _i64, _i64, _vp, _int, _f32, _f32, _vp, _vp]),
    "asq_linear_fp8_grouped": (_int, [_vp, _vp, _vp, _int, _vp, _int, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "asq_cast_e5m2": (_int, [_vp, _int, _vp, _i64, _vp]),
    "asq_quantize_mxfp8": (_int, [_vp, _int, _vp, _vp, _i64, _i64, _vp]),
    "asq_linear_mxfp8": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _vp, _vp]),
}


def lib():
    """Load libasq_hip.so once.  Fails loudly: there is no fallback implementation."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C autosmoothquant_amd/csrc`.  autosmoothquant_amd has no CPU/eager fallback.")
                # torch first: its wheel ships its own libamdhip64.so (soname libamdhip64.so.7, requested by libtorch_hip as "libamdhip64.so"),
                # libasq_hip.so asks for "libamdhip64.so.7".  Loaded after torch, this library binds to the runtime already in the process; loaded
                # BEFORE torch it would pull /opt/rocm's copy and torch a second one -- two HIP runtimes, and the one that initialises second finds
                # no device (hipError 100; seen with `python __graft_entry__.py smoke`, where build() used to load the library before smoke()
                # imported torch).  The host side above the C-ABI is torch plumbing anyway (ops.py).
                import torch  # noqa: F401
                h = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(h, name)  # AttributeError if the .so lacks a declared symbol
                    fn.restype, fn.argtypes = res, args
                if h.asq_version() != ASQ_VERSION:   # a stale build: argument lists and the workspace contract may differ
                    raise RuntimeError(f"{LIB_PATH} reports C-ABI version {h.asq_version()}, this package needs {ASQ_VERSION}: rebuild it (make -C autosmoothquant_amd/csrc)")
                _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().asq_last_error().decode("utf-8", "replace")
        if rc < 0:
            raise ValueError(f"{what or 'libasq_hip'}: {msg} (code {rc})")
        raise RuntimeError(f"{what or 'libasq_hip'}: {msg} (hipError {rc})")
