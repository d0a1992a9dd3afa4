"""ctypes binding of librubikhip.so (include/rubikhip.h) over PyTorch-ROCm tensors.

PyTorch is plumbing only: it owns the device buffers and the HIP stream; every cube
operation is a kernel of librubikhip.so.  There is NO fallback: if the library is missing
or was not built, importing this module's `lib()` raises, loudly.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch  # must be imported before the library so both share one HIP runtime (libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RUBIKHIP_LIB") or os.path.join(_HERE, "librubikhip.so")   # env override: A/B builds in experiments

FMT_NONE, FMT_CODE, FMT_U8, FMT_F16, FMT_F32, FMT_BF16 = 0, 1, 2, 3, 4, 5
_FMT_DTYPE = {FMT_U8: torch.uint8, FMT_F16: torch.float16, FMT_F32: torch.float32, FMT_BF16: torch.bfloat16}
STATUS_BAD_ACTION = 1

_lock = threading.Lock()
_lib = None
_inited = set()


class RubikHipError(RuntimeError):
    pass


def _declare(L):
    vp, i64, i32, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint64
    L.rc_version.restype = i32
    L.rc_build_id.restype = ctypes.c_char_p
    L.rc_last_error.restype = ctypes.c_char_p
    L.rc_init.argtypes = [i32]
    L.rc_get_tables.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
    L.rc_fill_solved.argtypes = [vp, i64, i64, i32, vp]
    L.rc_apply_moves.argtypes = [vp, vp, vp, i64, i64, i64, i32, vp, vp, vp, i32, i64, vp]
    L.rc_apply_moves_ex.argtypes = [vp, vp, vp, i64, i64, i64, i32, vp, vp, vp, i32, i64, vp, i32]
    L.rc_apply_moves_ws.argtypes = [vp, vp, vp, i64, i64, i64, i32, vp, vp, vp, i32, i64, vp, i64, vp]
    L.rc_encode_ws.argtypes = [vp, i64, i64, i32, vp, i32, i64, vp, i64, vp]
    L.rc_workspace_bytes.argtypes = [i32, i32, i64, i32]
    L.rc_workspace_bytes.restype = i64
    L.rc_facade_step.argtypes = [vp, i64, i32, i32, vp, ctypes.c_uint32, i32, vp]
    L.rc_facade_steps.argtypes = [vp, i64, i32, vp, i32, vp, ctypes.c_uint32, i32, vp]
    L.rc_facade_expand.argtypes = [vp, i64, i32, vp, ctypes.c_uint32, i32, i32, vp]
    L.rc_scramble.argtypes = [vp, i64, i64, i32, i32, u64, u64, i64, vp, vp, i64, vp, vp, vp]
    L.rc_legacy_scramble_actions.argtypes = [vp, vp, i32, i32, i64, i32, vp, i64, vp]
    L.rc_is_solved.argtypes = [vp, i64, i64, i32, vp, vp, vp]
    L.rc_encode.argtypes = [vp, i64, i64, i32, vp, i32, i64, vp]
    L.rc_onehot_from_code.argtypes = [vp, i64, i64, i32, vp, i32, vp]
    L.rc_onehot_from_code_ex.argtypes = [vp, i64, i64, i32, vp, i32, vp, i32]
    L.rc_onehot_from_code_blocks.argtypes = [vp, i64, i64, i32, vp, i32, i32, i64, i64, vp]
    L.rc_expand_children.argtypes = [vp, i64, i64, i32, vp, vp, vp, i64, vp]
    L.rc_expand_children_ex.argtypes = [vp, i64, i64, i32, vp, vp, vp, i64, vp, i32]
    L.rc_adi_generate.argtypes = [u64, u64, i64, i64, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rc_adi_generate_ex.argtypes = [u64, u64, i64, i64, i32, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32]
    L.rc_adi_generate_family.argtypes = [u64, u64, i64, i64, i32, i32, i64, vp, vp, vp, vp, vp, vp, i32]
    L.rc_family_layout.argtypes = [i32, vp, vp]
    L.rc_onehot_from_family.argtypes = [vp, i64, i64, i32, vp, i32, i64, vp]
    L.rc_adi_targets.argtypes = [vp, vp, vp, vp, i64, i64, i32, vp, vp, vp, vp]
    L.rc_adi_targets_depths.argtypes = [vp, i64, i64, vp, i64, vp, i64, vp, i64, i32, i32, vp, vp, vp, i64, vp]
    L.rc_onehot_from_family_depths.argtypes = [vp, i64, i64, i32, vp, i32, i64, i32, vp]
    L.rc_legacy_scramble_actions_ex.argtypes = [vp, vp, i32, i32, i64, i32, vp, i64, vp, i32]
    L.rc_read_status.argtypes = [vp, vp]
    L.rc_describe_dispatch.argtypes = [i32, i32, i64, i32, ctypes.c_uint32, i32, i32, ctypes.c_char_p, i32]
    L.rc_facade_release.argtypes = [vp]
    L.rc_host_alias.argtypes = [vp, vp]
    L.rc_scramble_from.argtypes = [vp, vp, i64, i64, i32, i32, u64, u64, i64, vp, vp, i64, vp, vp, vp]
    L.rc_search_pack.argtypes = [vp, vp, vp, i64, i64, i32, vp, vp, vp, vp]
    for name in ("rc_init", "rc_get_tables", "rc_fill_solved", "rc_apply_moves", "rc_apply_moves_ex", "rc_facade_step", "rc_facade_steps", "rc_facade_expand", "rc_scramble",
                 "rc_legacy_scramble_actions", "rc_is_solved", "rc_encode", "rc_onehot_from_code", "rc_expand_children",
                 "rc_expand_children_ex", "rc_adi_generate", "rc_adi_generate_ex", "rc_adi_targets", "rc_read_status",
                 "rc_describe_dispatch", "rc_facade_release", "rc_onehot_from_code_ex", "rc_apply_moves_ws", "rc_encode_ws", "rc_adi_generate_family", "rc_family_layout",
                 "rc_onehot_from_family", "rc_adi_targets_depths", "rc_onehot_from_family_depths", "rc_legacy_scramble_actions_ex", "rc_host_alias", "rc_scramble_from", "rc_search_pack", "rc_onehot_from_code_blocks"):
        getattr(L, name).restype = i32


def lib():
    """The loaded library (loads on first use).  Raises if it was not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RubikHipError(
                        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
                L = ctypes.CDLL(LIB_PATH)
                if not hasattr(L, "rc_build_id"):
                    raise RubikHipError(f"{LIB_PATH} predates rc_build_id (ABI < 600): rebuild it with __graft_entry__.build()")
                _declare(L)
                from . import _build
                try:                                                 # the binary must be the tree's sources (RC_ALLOW_STALE=1: A/B experiments)
                    _build.check_loaded(LIB_PATH, L.rc_build_id().decode(), _build.HIP_SOURCES)
                except RuntimeError as e:
                    raise RubikHipError(str(e)) from None
                _lib = L
    return _lib


def build_id() -> str:
    """The source hash the loaded library was built from (rc_build_id)."""
    return lib().rc_build_id().decode()


def check(rc):
    if rc != 0:
        raise RubikHipError(f"librubikhip error {rc}: {lib().rc_last_error().decode()}")


def init(device: torch.device):
    """rc_init for the tensor's device (once per device per process)."""
    if device.type != "cuda":
        raise RubikHipError(f"cube tensors must live on a HIP device, got {device}")
    cur = torch.cuda.current_device()
    idx = device.index if device.index is not None else cur
    if idx != cur:
        # kernels are launched for the CURRENT device; one process per GPU is the deployment model
        raise RubikHipError(f"tensor lives on cuda:{idx} but cuda:{cur} is current: call torch.cuda.set_device({idx}) "
                            f"(or use `with torch.cuda.device({idx}):`) around cube operations")
    if idx not in _inited:
        check(lib().rc_init(idx))
        _inited.add(idx)
    return idx


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the current stream's handle without building a Stream object


def stream_ptr(device=None):
    """hipStream_t of torch's CURRENT stream on `device` as a void* (looked up per call: the library is stream-ordered)."""
    if _raw_stream is not None:
        if isinstance(device, str):
            device = torch.device(device)
        idx = device.index if isinstance(device, torch.device) else device if isinstance(device, int) else None
        if idx is None:
            idx = torch.cuda.current_device()
        return ctypes.c_void_p(_raw_stream(idx))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def host_alias(t):
    """Device address of a pinned (host-mapped) CPU tensor, as a void*: kernels may read it in place (rc_host_alias)."""
    if t.is_cuda or not t.is_pinned():
        raise RubikHipError("host_alias: need a pinned CPU tensor")
    out = ctypes.c_void_p(0)
    check(lib().rc_host_alias(ctypes.c_void_p(t.data_ptr()), ctypes.byref(out)))
    return out


def pitch_for(n: int, align: int = 256) -> int:
    """Row pitch: >= n, multiple of `align` bytes (256 keeps every row segment line-aligned)."""
    return max(align, (n + align - 1) // align * align)


def dense_dtype(fmt):
    return _FMT_DTYPE[fmt]


def fmt_of(dtype):
    for f, d in _FMT_DTYPE.items():
        if d == dtype:
            return f
    raise RubikHipError(f"no dense one-hot format for dtype {dtype}")


def read_status(device=None) -> int:
    out = ctypes.c_uint32(0)
    check(lib().rc_read_status(ctypes.byref(out), stream_ptr(device)))
    return out.value


OP_STEP, OP_EXPAND, OP_ADI, OP_CODE_TO_DENSE, OP_FAMILY_TO_DENSE = 1, 2, 3, 4, 5
OUT_STATES, OUT_CODE, OUT_FLAGS, OUT_REWARD, OUT_INPLACE, OUT_DONE, OUT_WORKSPACE, OUT_FAMILY = 1, 2, 4, 8, 16, 32, 64, 128


def describe(op, cube_size, n, depth=0, outputs=0, fmt=FMT_NONE, variant=0) -> str:
    """The kernel instantiation + launch geometry a call WOULD use (rc_describe_dispatch): the library's own dispatch
    functions decide, nothing is launched.  Works without a GPU."""
    buf = ctypes.create_string_buffer(160)
    check(lib().rc_describe_dispatch(op, cube_size, n, depth, outputs, fmt, variant, buf, len(buf)))
    return buf.value.decode()


def get_tables(cube_size):
    """Host copy of the tables baked into the library (for tests: same numbers as tables.py)."""
    import numpy as np

    dims = (ctypes.c_int32 * 6)()
    check(lib().rc_get_tables(cube_size, None, None, None, None, None, None, dims))
    S, A, NC, NE, R, C = list(dims)
    out = dict(perm=np.zeros((A, S), np.uint8), solved=np.zeros(S, np.uint8), corner_defs=np.zeros((NC, 3), np.uint8),
               edge_defs=np.zeros((max(NE, 1), 2), np.uint8), corner_code=np.zeros(72, np.uint8), edge_code=np.zeros(72, np.uint8))
    check(lib().rc_get_tables(cube_size, *(v.ctypes.data_as(ctypes.c_void_p) for v in out.values()), dims))
    out["edge_defs"] = out["edge_defs"][:NE]
    out["dims"] = (S, A, NC, NE, R, C)
    return out


_family = {}


def family_layout(cube_size):
    """(NF, rows) of the FAMILY record (include/rubikhip.h rc_adi_generate_family): rows[a][p] = the family row that is slot p's code of
    child a (a = A: the parent), uint8 numpy [A + 1, SLOTS].  Needs no GPU."""
    if cube_size not in _family:
        import numpy as np
        A, SL = (12, 20) if cube_size == 3 else (6, 7)
        rows = np.zeros((A + 1, SL), np.uint8)
        nf = ctypes.c_int32(0)
        check(lib().rc_family_layout(cube_size, rows.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nf)))
        _family[cube_size] = (int(nf.value), rows)
    return _family[cube_size]
