"""ctypes binding of ``libanyv2v_hip.so`` (C ABI declared in ``include/anyv2v_hip.h``).

The HIP library IS the product: there is no eager / CPU fallback.  Importing this module when the
shared object is missing raises ``HipExtensionMissing`` with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libanyv2v_hip.so")
ABI_VERSION = 103   # ANYV2V_ABI_VERSION of include/anyv2v_hip.h the structures below mirror


class HipExtensionMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A0", C.c_void_p), ("A1", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("R", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("C0", C.c_int32), ("C1", C.c_int32),
        ("lda0", C.c_int32), ("lda1", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
        ("ldrv", C.c_int32), ("rowvec_div", C.c_int32), ("mode", C.c_int32),
        ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("stride", C.c_int32), ("up", C.c_int32), ("asym", C.c_int32), ("F", C.c_int32), ("HW", C.c_int32),
        ("act", C.c_int32), ("flags", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("ln_c1", C.c_void_p), ("ln_eps", C.c_float), ("reserved0", C.c_int32),
    ]


class FFDesc(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("R", C.c_void_p),
        ("Y", C.c_void_p), ("M", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("ldx", C.c_int32), ("ldr", C.c_int32),
        ("ldy", C.c_int32), ("flags", C.c_int32), ("reserved0", C.c_int32),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
        ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
        ("batch", C.c_int32), ("heads", C.c_int32), ("Sq", C.c_int32), ("Sk", C.c_int32),
        ("inner", C.c_int32),
        ("q_outer", C.c_int64), ("q_inner", C.c_int64), ("q_seq", C.c_int64),
        ("kv_outer", C.c_int64), ("kv_inner", C.c_int64), ("kv_seq", C.c_int64),
        ("kv_div", C.c_int32), ("qk_mod", C.c_int32), ("scale", C.c_float), ("flags", C.c_int32),
    ]


# every symbol include/anyv2v_hip.h declares: name -> (restype, argtypes)
_VP, _I32, _I64, _F32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SYMBOLS = {
    "anyv2v_gemm_f16": (C.c_int, [C.POINTER(GemmDesc), _VP]),
    "anyv2v_ff_geglu_f16": (C.c_int, [C.POINTER(FFDesc), _VP]),
    "anyv2v_groupnorm_f16": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _F32, _I32, _VP]),
    "anyv2v_groupnorm_scratch_floats": (C.c_int64, [_I32, _I32, _I32]),
    "anyv2v_groupnorm_partial_floats": (C.c_int64, [_I32, _I32, _I32, _I32]),
    "anyv2v_groupnorm_partial_f16": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _I32, _I32, _I32, _VP]),
    "anyv2v_groupnorm_apply_f16": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _F32, _I32, _I32, _VP]),
    "anyv2v_layernorm_f16": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _F32, _VP]),
    "anyv2v_attention_f16": (C.c_int, [C.POINTER(AttnDesc), _VP]),
    "anyv2v_attention_small_f16": (C.c_int, [C.POINTER(AttnDesc), _I32, _VP]),
    "anyv2v_attention_bias_f16": (C.c_int, [C.POINTER(AttnDesc), _I32, _VP, _VP]),
    "anyv2v_softmax_rows_f32_f16": (C.c_int, [_VP, _I32, _VP, _I32, _I32, _I32, _F32, _VP]),
    "anyv2v_silu_f16": (C.c_int, [_VP, _VP, _I64, _VP]),
    "anyv2v_add_f16": (C.c_int, [_VP, _VP, _VP, _I64, _VP]),
    "anyv2v_timestep_embedding_f16": (C.c_int, [_VP, _VP, _I32, _I32, _VP]),
    "anyv2v_ncfhw_to_tokens_f16": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "anyv2v_tokens_to_ncfhw_f16": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "anyv2v_adaptive_avgpool_f16": (C.c_int, [_VP, _VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP]),
    "anyv2v_copy_cols_f16": (C.c_int, [_VP, _I32, _I32, _VP, _I32, _I32, _I64, _I32, _VP]),
    "anyv2v_gather_rows_f16": (C.c_int, [_VP, _I32, _I32, _VP, _VP, _I32, _I32, _I64, _I32, _VP]),
    "anyv2v_rotary_f16": (C.c_int, [_VP, _I32, _I64, _I32, _I32, _I32, _I32, _I32, _I32, C.c_float, _VP]),
    "anyv2v_cfg_ddim_step_f16": (C.c_int, [_VP, _I32, _I32, _I32, _F32, _VP, _VP, _VP, _I32, _I32, _I32, _VP]),
    "anyv2v_ddim_step_f16": (C.c_int, [_VP, _VP, _VP, _F32, _F32, _F32, _F32, _I64, _VP]),
    "anyv2v_guided_step_f16": (C.c_int, [_VP, _I64, _I32, _I32, _I32, _F32, _F32, _I32, _F32, _F32, _F32, _F32, _VP, _VP, _VP]),
    "anyv2v_guided_step_noise_f16": (C.c_int, [_VP, _I64, _I32, _I32, _I32, _F32, _F32, _I32, _F32, _F32, _F32, _F32, _VP, _VP, _VP, _F32, _VP]),
    "anyv2v_set_batch_hint": (C.c_int, [_I32, _I32]),
    "anyv2v_last_error": (C.c_char_p, []),
    "anyv2v_version": (C.c_int, []),
    "anyv2v_selftest": (C.c_int, [_VP, _I64, _VP]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: the HIP kernels are the product path and there is no fallback. "
            f"Build with `make -C {os.path.join(_HERE, 'csrc')}` (or `python -c 'import __graft_entry__ as g; g.build()'`).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.anyv2v_version() < ABI_VERSION:   # descriptors carry no size field: a stale library would read past a shorter struct
        raise HipExtensionMissing(f"{LIB_PATH} is ABI {lib.anyv2v_version()}, this binding needs >= {ABI_VERSION}: rebuild it "
                                  f"(`make -C {os.path.join(_HERE, 'csrc')}`)")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().anyv2v_last_error()
        raise HipKernelError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
