"""ctypes binding of libx3dhip.so (include/x3dhip.h).

The library is mandatory for the product path: ``lib()`` raises if it is missing or if its
ABI version differs -- there is no CPU or eager-PyTorch fallback behind these ops.
Tensors are passed as ``tensor.data_ptr()``; the stream as
``torch.cuda.current_stream().cuda_stream``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libx3dhip.so")
ABI_VERSION = 7

ACT_NONE, ACT_RELU, ACT_SWISH = 0, 1, 2

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_Z = ctypes.c_size_t

# name -> (restype, argtypes).  Every symbol include/x3dhip.h declares is listed here;
# tests/test_abi.py checks the two against each other.
SIGNATURES = {
    "x3d_abi_version": (_I, []),
    "x3d_dw333_fwd_stats": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _I, _P, _I, _P]),
    "x3d_bn_stats_add_relu_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _I, _I, _P]),
    "x3d_clip_job_bytes": (_Z, []),
    "x3d_clip_preprocess": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "x3d_last_error": (ctypes.c_char_p, []),
    "x3d_last_kernel": (ctypes.c_char_p, []),
    "x3d_debug_poison_lds": (_I, [_P, _P]),
    "x3d_set_option": (_I, [ctypes.c_char_p, _I]),
    "x3d_get_option": (_I, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
    "x3d_reset_options": (_I, []),
    "x3d_option_count": (_I, []),
    "x3d_option_name": (ctypes.c_char_p, [_I]),
    "x3d_pw_tiles": (_I, [_I, _I, _I, _I, _I]),
    "x3d_pw_fwd_tiles": (_I, [_I, _I, _I, _I, _I, _I]),
    "x3d_pw_bwd_tiles": (_I, [_I, _I, _I, _I, _I, _I]),
    "x3d_pw_wants_packed": (_I, [_I, _I]),
    "x3d_pw_pack_floats": (_Z, [_I, _I, _I]),
    "x3d_pw_pack_items": (_Z, [_I, _I, _I]),
    "x3d_pw_pack": (_I, [_P, _P, _I, _I, _I, _P]),
    "x3d_pw_pack_job_bytes": (_Z, []),
    "x3d_pw_pack_batch": (_I, [_P, _P, _I, _P]),
    "x3d_pw_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P]),
    "x3d_pw_bwd_data": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I, _P]),
    "x3d_pw_bwd_data_res": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P]),
    "x3d_pw_bwd_fused_ok": (_I, [_I, _I, _I, _I, _I, _I]),
    "x3d_pw_bwd_fused_groups": (_I, [_I, _I]),
    "x3d_pw_bwd_fused_tiles": (_I, [_I, _I]),
    "x3d_pw_bwd_fused": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "x3d_pw_wgrad_groups": (_I, [_I, _I, _I, _I, _I]),
    "x3d_pw_bwd_weight": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "x3d_reduce_partials": (_I, [_P, _P, _I, _I, _P]),
    "x3d_wgrad_job_bytes": (ctypes.c_size_t, []),
    "x3d_pw_bwd_weight_batch": (_I, [_P, _I, _P]),
    "x3d_reduce_partials_batch": (_I, [_P, _P, _P, _P, _I, _P]),
    "x3d_dw_tiles": (_I, [_I, _I, _I, _I, _I]),
    "x3d_dw333_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P]),
    "x3d_dw_bwd_tiles": (_I, [_I, _I, _I, _I, _I, _I]),
    "x3d_dw333_bwd_stats": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "x3d_dw333_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "x3d_stem133_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "x3d_stem_wgrad_groups": (_I, [_I, _I]),
    "x3d_stem133_bwd_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "x3d_dw5t_tiles": (_I, [_I]),
    "x3d_dw5t_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "x3d_dw5t_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "x3d_finalize_scratch_bytes": (_Z, [_I, _I, _I]),
    "x3d_bn_fwd_finalize": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "x3d_bn_eval_coef": (_I, [_P, _P, _P, _P, _F, _I, _I, _P, _P]),
    "x3d_se_fwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "x3d_se_bn_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "x3d_bn_bwd_finalize": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P]),
    "x3d_se_bn_bwd_finalize": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P,
                                    _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "x3d_ew_tiles": (_I, [_I]),
    "x3d_bn_add_relu_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "x3d_bn_add_relu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "x3d_bn_relu_pool_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "x3d_bn_relu_pool_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "x3d_head_scratch_floats": (_Z, [_I, _I, _I, _I]),
    "x3d_head_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P]),
    "x3d_head_ce": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P]),
    "x3d_head_advance_rng": (_I, [_P, _P, _P]),
    "x3d_head_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "x3d_loc_losses": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "x3d_bn_rowstats": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "x3d_bn_affine": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "x3d_grad_accumulate": (_I, [_P, _P, _Z, _F, _I, _P]),
    "x3d_sgd_fused": (_I, [_P, _P, _P, _Z, _F, _F, _F, _F, _I, _P]),
}

_lib = None


class X3DHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises X3DHipError when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise X3DHipError(
            "libx3dhip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C x3d-multigrid_amd/csrc`). The X3D product path has no fallback." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64: import torch FIRST, so that the library's HIP dependency resolves to the runtime
    # torch already loaded.  Loaded the other way round (this library before torch, e.g. `python __graft_entry__.py smoke`,
    # which builds and then smokes in one process) the process holds two HIP runtimes and the library's launches fail with
    # "no ROCm-capable device is detected".
    import torch  # noqa: F401
    h = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise X3DHipError("libx3dhip.so lacks symbol %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    v = h.x3d_abi_version()
    if v != ABI_VERSION:
        raise X3DHipError("libx3dhip.so ABI %d != expected %d" % (v, ABI_VERSION))
    _lib = h
    return h


def set_option(name, value):
    """Library option `name` := value (include/x3dhip.h x3d_set_option); returns the previous value."""
    prev = get_option(name)
    check(lib().x3d_set_option(name.encode(), int(value)))
    return prev


def get_option(name):
    v = ctypes.c_int(0)
    check(lib().x3d_get_option(name.encode(), ctypes.byref(v)))
    return v.value


def last_kernel():
    """Kernel template name the last pointwise / channelwise entry point of this thread launched (x3d_last_kernel)."""
    return lib().x3d_last_kernel().decode()


def option_names():
    L = lib()
    return [L.x3d_option_name(i).decode() for i in range(L.x3d_option_count())]


class options:
    """Context manager: `with _lib.options(fb_grid=8, dgrad_f32=1): ...` -- options restored on exit."""

    def __init__(self, **kw):
        self.kw, self.prev = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.prev[k] = set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            set_option(k, v)
        return False


def check(rc):
    if rc != 0:
        raise X3DHipError("libx3dhip: error %d: %s" % (rc, lib().x3d_last_error().decode("utf-8", "replace")))


def ptr(t):
    """data_ptr of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
