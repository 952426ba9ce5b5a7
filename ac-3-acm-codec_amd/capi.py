"""ctypes binding of libac3mi.so — signatures follow include/ac3mi.h one to one."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AC3MI_LIB") or os.path.join(_HERE, "libac3mi.so")   # override: A/B builds only
HEADER_DIR = os.path.join(os.path.dirname(_HERE), "include")

c_void_p, c_int, c_float, c_size_t, c_char_p = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_size_t, ctypes.c_char_p)


class AC3MIError(RuntimeError):
    pass


class XformDescC(ctypes.Structure):
    _fields_ = [("acmod", c_int), ("lfeon", c_int), ("output", c_int), ("bias", c_float)]


class DecodeDescC(ctypes.Structure):
    _fields_ = [("flags", c_int), ("level", c_float), ("bias", c_float), ("dynrng", c_int),
                ("acmod", c_int), ("lfeon", c_int), ("frame_bytes", c_int)]


class DecodeTapsC(ctypes.Structure):
    _fields_ = [("d_coef", c_void_p), ("d_blksw", c_void_p), ("d_exp", c_void_p), ("d_bap", c_void_p),
                ("d_dynrng_out", c_void_p), ("d_dynrng_in", c_void_p)]


class EncodeDescC(ctypes.Structure):
    _fields_ = [("sample_rate", c_int), ("bit_rate", c_int), ("channels", c_int)]


class EncodeTapsC(ctypes.Structure):
    _fields_ = [("d_mdct", c_void_p), ("d_exponent", c_void_p), ("d_exp_samples", c_void_p),
                ("d_encoded_exp", c_void_p), ("d_bap", c_void_p), ("d_exp_strategy", c_void_p),
                ("d_snroffst", c_void_p)]


_lib = None


def declared_symbols():
    """Every function name declared in include/*.h (the drop-in boundary)."""
    names = []
    for fn in sorted(os.listdir(HEADER_DIR)):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(HEADER_DIR, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"typedef[^;{]*\(\s*\*[^;]*;", "", text)        # function-pointer typedefs
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text):
            if m.group(1) not in ("defined", "sizeof", "void", "int"):
                names.append(m.group(1))
        for m in re.finditer(r"extern\s+\w+\s+(\w+)\s*\[", text):      # exported tables (MapTab)
            names.append(m.group(1))
    return sorted(set(names))


def load_library():
    """Load libac3mi.so; raises loudly when the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AC3MIError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback)" % LIB_PATH)
    # torch wheels carry their own libamdhip64; if libac3mi.so pulled in /opt/rocm's copy
    # first, torch would later see "No HIP GPUs".  Load torch's runtime first so the
    # process holds exactly one HIP runtime (plumbing concern only: C hosts have no torch).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.ac3mi_create.restype = c_void_p
    lib.ac3mi_create.argtypes = [c_int]
    lib.ac3mi_destroy.argtypes = [c_void_p]
    lib.ac3mi_destroy.restype = None
    lib.ac3mi_last_error.restype = c_char_p
    lib.ac3mi_last_error.argtypes = [c_void_p]
    lib.ac3mi_device_count.restype = c_int
    lib.ac3mi_dev_alloc.restype = c_void_p
    lib.ac3mi_dev_alloc.argtypes = [c_void_p, c_size_t]
    lib.ac3mi_dev_free.argtypes = [c_void_p, c_void_p]
    lib.ac3mi_dev_free.restype = None
    lib.ac3mi_memcpy_h2d.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
    lib.ac3mi_memcpy_d2h.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
    lib.ac3mi_memset.argtypes = [c_void_p, c_void_p, c_int, c_size_t]
    lib.ac3mi_set_encode_mode.argtypes = [c_void_p, c_int]
    lib.ac3mi_memcpy_d2d.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
    lib.ac3mi_sync.argtypes = [c_void_p]
    lib.ac3mi_timer_start.argtypes = [c_void_p]
    lib.ac3mi_timer_stop.argtypes = [c_void_p, ctypes.POINTER(c_float)]
    lib.ac3mi_xform_planes.argtypes = [ctypes.POINTER(XformDescC), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    lib.ac3mi_imdct_batch.argtypes = [c_void_p, ctypes.POINTER(XformDescC), c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int, c_int]
    lib.ac3mi_syncinfo.argtypes = [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    lib.ac3mi_decode_planes.argtypes = [ctypes.POINTER(DecodeDescC), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    lib.ac3mi_decode_batch.argtypes = [c_void_p, ctypes.POINTER(DecodeDescC), c_void_p, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(DecodeTapsC)]
    lib.ac3mi_decode_s16_batch.argtypes = [c_void_p, ctypes.POINTER(DecodeDescC), c_void_p, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p, c_void_p]
    lib.ac3mi_encode_frame_bytes.argtypes = [ctypes.POINTER(EncodeDescC)]
    lib.ac3mi_encode_tables.argtypes = [c_void_p] * 5
    lib.ac3mi_encode_batch.argtypes = [c_void_p, ctypes.POINTER(EncodeDescC), c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_int, c_int, ctypes.POINTER(EncodeTapsC)]
    lib.ac3mi_set_decode_mode.argtypes = [c_void_p, ctypes.c_int]
    lib.ac3mi_set_mix_state.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.ac3mi_probe_valu_rate.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_double)]
    lib.ac3mi_probe_salu_rate.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_double)]
    lib.ac3mi_probe_mixed_rate.argtypes = [c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    lib.ac3mi_probe_copy_rate.argtypes = [c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double)]
    lib.ac3mi_set_tile_frames.argtypes = [c_void_p, ctypes.c_longlong]
    lib.ac3mi_transcode_batch.argtypes = [c_void_p, ctypes.POINTER(DecodeDescC), ctypes.POINTER(EncodeDescC), c_void_p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          ctypes.c_int, c_void_p]
    _lib = lib
    return lib
