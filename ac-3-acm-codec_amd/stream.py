"""ctypes mirror of include/ac3mi_stream.h — the ACM stream messages (open / size / convert / close) of the
reference driver (src/AC3ACM.cpp:1430-1628, 1665-1798, 1862-2131, 2139-2363) over the batched engine."""
import ctypes

from . import capi

WAVE_FORMAT_PCM, WAVE_FORMAT_AC3, WAVE_FORMAT_EXTENSIBLE = 0x0001, 0x2000, 0xFFFE
ACM_MULTICHANNEL, ACM_DYNAMICRANGE, ACM_DOLBYSURROUND, ACM_NOEXTENSIBLE = 1, 2, 4, 32
MMSYSERR_NOERROR, MMSYSERR_NOMEM, MMSYSERR_NOTSUPPORTED, MMSYSERR_INVALPARAM, ACMERR_NOTPOSSIBLE = 0, 7, 8, 11, 512
STREAMCONVERTF_START = 0x10
STREAMSIZEF_SOURCE, STREAMSIZEF_DESTINATION = 0, 1


class WaveFmt(ctypes.Structure):
    _fields_ = [("format_tag", ctypes.c_uint16), ("channels", ctypes.c_uint16), ("samples_per_sec", ctypes.c_uint32),
                ("avg_bytes_per_sec", ctypes.c_uint32), ("block_align", ctypes.c_uint16),
                ("bits_per_sample", ctypes.c_uint16), ("channel_mask", ctypes.c_uint32)]


class StreamHeader(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("src_len", ctypes.c_uint32), ("src_used", ctypes.c_uint32),
                ("dst", ctypes.c_void_p), ("dst_len", ctypes.c_uint32), ("dst_used", ctypes.c_uint32),
                ("flags", ctypes.c_uint32)]


def pcm_format(channels, rate, extensible=False):
    masks = (0x4, 0x3, 0x7, 0x33, 0x37, 0x3F)
    return WaveFmt(WAVE_FORMAT_EXTENSIBLE if extensible else WAVE_FORMAT_PCM, channels, rate, channels * 2 * rate,
                   channels * 2, 16, masks[channels - 1] if extensible else 0)


def ac3_format(channels, rate, kbps, block_align=None):
    return WaveFmt(WAVE_FORMAT_AC3, channels, rate, kbps * 125, block_align if block_align is not None else 1, 0, 0)


def _lib():
    lib = capi.load_library()
    if not getattr(lib, "_stream_bound", False):
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        lib.ac3mi_pool_create.restype = vp
        lib.ac3mi_pool_create.argtypes = [vp, ctypes.c_int]
        lib.ac3mi_pool_destroy.restype = None
        lib.ac3mi_pool_destroy.argtypes = [vp]
        lib.ac3mi_stream_open.argtypes = [vp, ctypes.POINTER(WaveFmt), ctypes.POINTER(WaveFmt), u32, ctypes.c_int,
                                          ctypes.POINTER(vp)]
        lib.ac3mi_stream_close.argtypes = [vp]
        lib.ac3mi_stream_size.argtypes = [vp, ctypes.c_int, u32, ctypes.POINTER(u32)]
        lib.ac3mi_stream_convert.argtypes = [vp, ctypes.POINTER(StreamHeader)]
        lib.ac3mi_stream_convert_many.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ctypes.POINTER(StreamHeader)),
                                                  ctypes.c_int]
        lib.ac3mi_stream_framesize.argtypes = [ctypes.POINTER(WaveFmt)]
        lib._stream_bound = True
    return lib


class Pool:
    def __init__(self, engine, capacity):
        self.lib = _lib()
        self.engine = engine
        self.handle = self.lib.ac3mi_pool_create(ctypes.c_void_p(engine.ctx), capacity)
        if not self.handle:
            raise RuntimeError("ac3mi_pool_create failed: " + self.lib.ac3mi_last_error(engine.ctx).decode())

    def close(self):
        if self.handle:
            self.lib.ac3mi_pool_destroy(ctypes.c_void_p(self.handle))
            self.handle = None

    def open(self, src, dst, driver_flags=ACM_MULTICHANNEL | ACM_DYNAMICRANGE, query=False):
        """-> (result code, Stream or None)"""
        h = ctypes.c_void_p()
        rc = self.lib.ac3mi_stream_open(ctypes.c_void_p(self.handle), ctypes.byref(src), ctypes.byref(dst), driver_flags,
                                        1 if query else 0, ctypes.byref(h))
        return rc, (Stream(self, h.value) if rc == 0 and h.value else None)

    def convert_many(self, streams, headers):
        n = len(streams)
        sarr = (ctypes.c_void_p * n)(*[s.handle for s in streams])
        harr = (ctypes.POINTER(StreamHeader) * n)(*[ctypes.pointer(h) for h in headers])
        return self.lib.ac3mi_stream_convert_many(sarr, harr, n)


class Stream:
    def __init__(self, pool, handle):
        self.pool, self.handle = pool, handle

    def close(self):
        if self.handle:
            self.pool.lib.ac3mi_stream_close(ctypes.c_void_p(self.handle))
            self.handle = None

    def size(self, query, nbytes):
        out = ctypes.c_uint32(0)
        rc = self.pool.lib.ac3mi_stream_size(ctypes.c_void_p(self.handle), query, nbytes, ctypes.byref(out))
        return rc, out.value

    def convert(self, header):
        return self.pool.lib.ac3mi_stream_convert(ctypes.c_void_p(self.handle), ctypes.byref(header))


def framesize(fmt):
    return _lib().ac3mi_stream_framesize(ctypes.byref(fmt))
