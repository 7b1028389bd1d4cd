"""ctypes binding of libmi3dgs.so (the C-ABI declared in include/mi3dgs.h).

There is NO fallback: if the HIP library is missing or a call fails this module raises.
The product path never imports anything from `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI3DGS_LIB: another build of the same ABI, for same-box A/B measurements of two versions of a kernel (announced on
# stderr when it is used: a leaked variable must not silently swap the library under a training job)
LIB_PATH = os.environ.get("MI3DGS_LIB") or os.path.join(_HERE, "libmi3dgs.so")
# the EXPERIMENTS build (rejected variants, timing experiments with wrong results): never loaded by the product path,
# only by experiments_lib() -- the A/B tools under tools/ and the tests that use a rejected-but-correct variant as a yardstick
EXP_LIB_PATH = os.path.join(_HERE, "libmi3dgs_exp.so")
CSRC_DIR = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
ABI_VERSION = 7

_lib: Optional[C.CDLL] = None
_exp_lib: Optional[C.CDLL] = None

_f = C.c_void_p      # device pointers travel as void*
_i = C.c_int
_ll = C.c_longlong
_fl = C.c_float
_sz = C.c_size_t
_u32 = C.c_uint32

_SIGNATURES = {
    "mi3dgs_last_error": (C.c_char_p, []),
    "mi3dgs_abi_version": (_i, []),
    "mi3dgs_splat_stride": (_i, []),
    "mi3dgs_grad_stride": (_i, []),
    "mi3dgs_profile_enable": (_i, [_i]),
    "mi3dgs_profile_read": (_sz, [C.c_char_p, _sz]),
    "mi3dgs_project_fwd": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _f, _i, _i, _f, _f, _i, _i, _fl, _fl, _fl, _fl, _i,
                                _f, _f, _f, _f]),
    "mi3dgs_project_bwd": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, _i, _i, _f, _f, _i, _i, _fl, _i, _f, _f, _f,
                                _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f]),
    "mi3dgs_project_bwd_adam": (_i, [_i, _f, _f, _f, _f, _f, _f, _i, _f, _f, _i, _i, _fl, _i, _f, _f, _f,
                                     C.POINTER(_f), C.POINTER(_f), C.POINTER(_fl), _i, _fl, _fl, _fl, _fl, _fl,
                                     _f, _f, _f, _i, _f]),
    "mi3dgs_project_bwd_adam_mcmc": (_i, [_i, _f, _f, _f, _f, _f, _f, _i, _f, _f, _i, _i, _fl, _i, _f, _f, _f,
                                          C.POINTER(_f), C.POINTER(_f), C.POINTER(_fl), _i, _fl, _fl, _fl, _fl, _fl, _f]),
    "mi3dgs_bin_workspace_bytes": (_sz, [_i, _i, _ll]),
    "mi3dgs_bin_count": (_i, [_i, _i, _f, _f, _i, _i, _i, _i, _i, _f, _f, _f, _sz, _ll, _f]),
    "mi3dgs_bin_emit": (_i, [_i, _i, _f, _f, _i, _i, _i, _i, _i, _f, _ll, _f, _f, _f, _f, _f, _sz, _f]),
    "mi3dgs_bin_tiles": (_i, [_i, _i, _f, _f, _i, _i, _i, _i, _i, _f, _ll, _f, _f, _f, _f, _f, _f, _f, _sz, _f]),
    "mi3dgs_sort_workspace_bytes": (_sz, [_ll]),
    "mi3dgs_sort_pairs_u32": (_i, [_f, _f, _ll, _i, _f, _sz, _f]),
    "mi3dgs_debug_set_sort_mode": (_i, [_i]),
    "mi3dgs_debug_set_raster_mode": (_i, [_i]),
    "mi3dgs_debug_set_raster_fwd_segments": (_i, [_i]),
    "mi3dgs_debug_set_emit_mode": (_i, [_i]),
    "mi3dgs_async_errors": (_i, [C.POINTER(_u32), _i]),
    "mi3dgs_scan_workspace_bytes": (_sz, [_ll]),
    "mi3dgs_scan_exclusive_u32": (_i, [_f, _f, _ll, _f, _f, _sz, _f]),
    "mi3dgs_rasterize_fwd": (_i, [_i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _sz, _f]),
    "mi3dgs_rasterize_bwd": (_i, [_i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _f, _ll, _f, _f, _sz, _f]),
    "mi3dgs_raster_seg_workspace_bytes": (_sz, [_i, _ll]),
    "mi3dgs_raster_seg_workspace_init": (_i, [_f, _sz, _f]),
    "mi3dgs_adam_culled_groups": (_i, [_i, _f, _f, _f, _f, _f, _i, _fl, _fl, _fl, _fl, _fl, _f]),
    "mi3dgs_loss_fwd": (_i, [_i, _i, _i, _f, _f, _f, _f, _f, _f, _f]),
    "mi3dgs_loss_bwd": (_i, [_i, _i, _i, _f, _f, _f, _f, _f, _fl, _fl, _f, _f]),
    "mi3dgs_loss_fwd_u8": (_i, [_i, _i, _i, _f, _f, _fl, _f, _f, _f, _f, _f]),
    "mi3dgs_loss_bwd_u8": (_i, [_i, _i, _i, _f, _f, _fl, _f, _f, _f, _fl, _fl, _f, _f]),
    "mi3dgs_scale_reg": (_i, [_i, _f, _fl, _fl, _f, _f, _f]),
    "mi3dgs_adam_step": (_i, [_i, C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), C.POINTER(_ll),
                              C.POINTER(_fl), _i, _fl, _fl, _fl, _fl, _f]),
    "mi3dgs_debug_hbm_stream": (_i, [_f, _ll, _i, _i, _f]),
    "mi3dgs_densify_decide": (_i, [_i, _f, _f, _f, _f, _f, _fl, _fl, _fl, _fl, _fl, _fl, _i, _i, _f, _f, _f]),
    "mi3dgs_densify_scatter": (_i, [_i, _ll, C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), C.POINTER(_f), C.POINTER(_f),
                                    C.POINTER(_f), _f, _f, _ll, _u32, _f, _f]),
    "mi3dgs_reset_opacity": (_i, [_i, _f, _fl, _f, _f, _f]),
    "mi3dgs_mcmc_relocation": (_i, [_i, _f, _f, _f, _f, _f, _f, _f]),
    "mi3dgs_mcmc_inject_noise": (_i, [_i, _f, _f, _f, _f, _fl, _u32, _f]),
    "mi3dgs_mcmc_regularise": (_i, [_i, _f, _f, _fl, _fl, _f, _f, _f]),
    "mi3dgs_knn_workspace_bytes": (_sz, [_ll]),
    "mi3dgs_knn": (_i, [_ll, _f, _i, _f, _f, _f, _sz, _f]),
    "mi3dgs_image_downscale_area": (_i, [_f, _i, _i, _i, _f, _i, _i, _i, _f]),
    "mi3dgs_image_u8_to_f32": (_i, [_f, _ll, _f, _fl, _f]),
    "mi3dgs_image_undistort": (_i, [_f, _i, _i, _i, _f, _i, _i, C.POINTER(_fl), C.POINTER(_fl), _i, C.POINTER(_fl), _i,
                                    _i, _f]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))


class Mi3dgsError(RuntimeError):
    pass


def _open(path: str) -> C.CDLL:
    handle = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)     # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    if handle.mi3dgs_abi_version() != ABI_VERSION:
        raise Mi3dgsError(f"{path}: ABI version mismatch")
    return handle


def lib() -> C.CDLL:
    """Load the library (once).  Raises if it has not been built: there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise Mi3dgsError(
                f"{LIB_PATH} not found: build it with `make -C {CSRC_DIR}` "
                "(or __graft_entry__.build()).  mi3dgs has no fallback path.")
        if os.environ.get("MI3DGS_LIB"):
            import sys
            print(f"[mi3dgs] WARNING: MI3DGS_LIB is set, loading {LIB_PATH} instead of the product library", file=sys.stderr, flush=True)
        _lib = _open(LIB_PATH)
    return _lib


def experiments_lib() -> C.CDLL:
    """The -DMI3DGS_EXPERIMENTS build of the same sources (A/B tools, yardstick tests).  Its own globals, its own device
    error word; calls go through `exp_call`."""
    global _exp_lib
    if _exp_lib is None:
        if not os.path.isfile(EXP_LIB_PATH):
            raise Mi3dgsError(f"{EXP_LIB_PATH} not found: build it with `make -C {CSRC_DIR}`")
        _exp_lib = _open(EXP_LIB_PATH)
    return _exp_lib


def exp_call(name: str, *args) -> None:
    h = experiments_lib()
    rc = getattr(h, name)(*args)
    if rc != 0:
        msg = h.mi3dgs_last_error()
        raise Mi3dgsError(msg.decode() if msg else f"mi3dgs (experiments) call failed with code {rc}")


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().mi3dgs_last_error()
        raise Mi3dgsError(msg.decode() if msg else f"mi3dgs call failed with code {rc}")


def profile_enable(on: bool) -> None:
    lib().mi3dgs_profile_enable(int(bool(on)))


def profile_read() -> dict:
    """{kernel tag: (launches, total_ms)} since the last read; waits for the kernels."""
    buf = C.create_string_buffer(1 << 16)
    n = lib().mi3dgs_profile_read(buf, len(buf))
    out = {}
    for line in buf.raw[:n].decode().splitlines():
        tag, cnt, ms = line.rsplit(" ", 2)
        out[tag] = (int(cnt), float(ms))
    return out


def async_errors(reset: bool = True) -> int:
    """Bits left by chained kernels whose bounded waits ran out (0 = fine).  Synchronises."""
    v = _u32(0)
    check(lib().mi3dgs_async_errors(C.byref(v), int(reset)))
    return int(v.value)


# Optional stage hook (used by bench.py to bracket each C-ABI call with HIP events on the
# launch stream).  hook(name, thunk) must call thunk() exactly once.
STAGE_HOOK = None


_SYNC_EACH_CALL = os.environ.get("MI3DGS_SYNC_EACH_CALL") == "1"      # debugging aid: a device fault surfaces in the call that caused it


def call(name: str, *args) -> None:
    fn = getattr(lib(), name)
    if _SYNC_EACH_CALL:
        import sys
        import torch
        check(fn(*args))
        print(f"[mi3dgs sync] {name}", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
    elif STAGE_HOOK is None:
        check(fn(*args))
    else:
        STAGE_HOOK(name, lambda: check(fn(*args)))
