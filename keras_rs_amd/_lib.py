"""ctypes binding of libkrs_hip.so (the C ABI of include/krs.h).

The HIP library is the product: there is no CPU fallback.  Importing this
module works anywhere (so `-m "not gpu"` tests can check the symbols), but any
compute call without the library or without a GPU raises.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KRS_LIB", os.path.join(_HERE, "libkrs_hip.so"))

F32, BF16 = 0, 1
I32, I64 = 0, 1
SUM, MEAN, SQRTN = 0, 1, 2
COMBINERS = {"sum": SUM, "mean": MEAN, "sqrtn": SQRTN}
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
FLAG_ID_OUT_OF_RANGE = 1
FLAG_BAD_OFFSETS = 2
FLAG_CAPACITY_OVERFLOW = 4

# struct layouts of include/krs.h
TABLE_DT = np.dtype(
    [("weights", "<u8"), ("slot", "<u8"), ("row_base", "<i8"), ("vocab", "<i4"), ("lr", "<f4")]
)
FEATURE_DT = np.dtype(
    [("ids_base", "<i8"), ("table", "<i4"), ("hot", "<i4"), ("combiner", "<i4"), ("out_col", "<i4")]
)
SHARD_FEATURE_DT = np.dtype(
    [("ids_base", "<i8"), ("comp_off", "<i8"), ("hot", "<i4"), ("combiner", "<i4"), ("vocab", "<i4"),
     ("reserved", "<i4")]
)


class GemmEpilogue(C.Structure):
    _fields_ = [
        ("bias", C.c_void_p),
        ("act", C.c_int32),
        ("diag_scale", C.c_float),
        ("x0", C.c_void_p),
        ("x", C.c_void_p),
        ("ldx", C.c_int64),
        ("u_out", C.c_void_p),
        ("ldu", C.c_int64),
        ("r", C.c_void_p),
        ("ldr", C.c_int64),
        ("beta", C.c_float),
    ]


# Every symbol include/krs.h declares (checked by tests/test_capi_symbols.py).
SYMBOLS = [
    "krs_version",
    "krs_last_error",
    "krs_embed_bag_fwd",
    "krs_embed_bag_bwd_workspace_bytes",
    "krs_embed_bag_bwd_plan",
    "krs_embed_bag_bwd_plan_tables",
    "krs_embed_bag_bwd_dense",
    "krs_embed_bag_bwd_fused_sgd",
    "krs_embed_bag_bwd_fused_adagrad",
    "krs_embed_bag_bwd_fused_adagrad_rowwise",
    "krs_embed_bag_bwd_fused_adam",
    "krs_embed_bag_bwd_fused_adam_dyn",
    "krs_store_f32",
    "krs_embed_bag_bwd_fused_ftrl",
    "krs_embed_bag_bwd_sparse",
    "krs_gemm",
    "krs_gemm_workspace_bytes",
    "krs_gemm_set_option",
    "krs_embed_set_option",
    "krs_cross_epilogue_fwd",
    "krs_cross_epilogue_bwd",
    "krs_gemm_cross_bwd",
    "krs_gemm_cross_bwd_workspace_bytes",
    "krs_colsum",
    "krs_colsum_workspace_bytes",
    "krs_cast_transpose",
    "krs_cast_transpose_many",
    "krs_dense_adagrad",
    "krs_dense_act_bwd",
    "krs_dot_interaction_fwd",
    "krs_dot_interaction_bwd",
    "krs_dot_interaction_bwd_accumulate",
    "krs_mod_bucketize_workspace_bytes",
    "krs_mod_bucketize",
    "krs_shard_route_workspace_bytes",
    "krs_shard_route",
    "krs_shard_unpack_workspace_bytes",
    "krs_shard_unpack",
    "krs_shard_combine",
    "krs_shard_static_block_words",
    "krs_shard_route_static",
    "krs_shard_unpack_static",
    "krs_publish_i64",
    "krs_bce_fwd_bwd",
]

_lib = None


class KrsError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Loads libkrs_hip.so; loud failure when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KrsError(
                f"{LIB_PATH} is missing: build it with `python -m keras_rs_amd.build` "
                "(hipcc --offload-arch=gfx950).  keras_rs_amd has no CPU fallback."
            )
        _lib = C.CDLL(LIB_PATH)
        _lib.krs_last_error.restype = C.c_char_p
        for name in ("krs_embed_bag_bwd_workspace_bytes", "krs_gemm_workspace_bytes",
                     "krs_mod_bucketize_workspace_bytes", "krs_shard_route_workspace_bytes",
                     "krs_shard_unpack_workspace_bytes", "krs_colsum_workspace_bytes",
                     "krs_gemm_cross_bwd_workspace_bytes"):
            getattr(_lib, name).restype = C.c_size_t
        _lib.krs_shard_static_block_words.restype = C.c_int64
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise KrsError(f"{what} failed ({rc}): {lib().krs_last_error().decode()}")


def require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise KrsError(
            f"{what}: tensor is on {t.device}; keras_rs_amd runs on MI355X HIP kernels only "
            "(no CPU fallback)"
        )


def ptr(t: torch.Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def fdtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise KrsError(f"unsupported float dtype {t.dtype} (float32 / bfloat16 only)")


def itype(t: torch.Tensor) -> int:
    if t.dtype == torch.int32:
        return I32
    if t.dtype == torch.int64:
        return I64
    raise KrsError(f"unsupported index dtype {t.dtype} (int32 / int64 only)")


# page-locked staging blocks set aside for descriptor uploads issued while a stream is CAPTURING (hipHostMalloc is not
# allowed then, and the memcpy node of the graph reads its source at every replay: a block handed out there is never
# reused)
_CAPTURE_PIN_BYTES = 4096
_capture_pins: list = []
_capture_pins_held: list = []


def _reserve_capture_pins(count: int = 64) -> None:
    if not _capture_pins and not _capture_pins_held:
        block = torch.empty(count * _CAPTURE_PIN_BYTES, dtype=torch.uint8).pin_memory()
        _capture_pins.extend(block[i * _CAPTURE_PIN_BYTES:(i + 1) * _CAPTURE_PIN_BYTES] for i in range(count))


def struct_to_device(arr: np.ndarray, device) -> torch.Tensor:
    """Uploads a numpy structured array (krs_table / krs_feature) as raw bytes."""
    host = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy())
    if torch.device(device).type == "cuda":
        if torch.cuda.is_current_stream_capturing():
            n = host.numel()
            if n > _CAPTURE_PIN_BYTES or not _capture_pins:
                raise KrsError("descriptor upload during graph capture: run the step eagerly first (descriptors are "
                               "cached; the few that change per step use a reserved page-locked block of 4 KB)")
            pin = _capture_pins.pop()
            _capture_pins_held.append(pin)
            pin[:n].copy_(host)
            dev = torch.empty(n, dtype=torch.uint8, device=device)
            dev.copy_(pin[:n], non_blocking=True)
            return dev
        _reserve_capture_pins()
        # page-locked staging + asynchronous copy: a pageable upload makes the host wait for everything
        # queued on the stream (a pipeline drain per descriptor; the host allocator keeps the staging
        # block alive until the copy has run)
        return host.pin_memory().to(device, non_blocking=True)
    return host.to(device)
