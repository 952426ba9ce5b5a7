"""Host-side mirror of the batched engine: torch tensors in, torch tensors out.

torch is plumbing here (device memory + dtype checks); every computation happens
in libac3mi.so.  Mirrors the stage boundaries of the reference:
``Engine.imdct_batch`` = the synthesis stage of a52_block (liba52/parse.c:881-937).
"""
import ctypes
from dataclasses import dataclass

from . import capi


@dataclass
class XformDesc:
    acmod: int = 7
    lfeon: int = 1
    output: int = 7 | 16
    bias: float = 0.0

    def c(self):
        return capi.XformDescC(self.acmod, self.lfeon, self.output, self.bias)


@dataclass
class DecodeDesc:
    """Arguments of a52_frame() plus the coded configuration a52_syncinfo() reported."""
    flags: int = 7 | 16
    level: float = 1.0
    bias: float = 0.0
    dynrng: int = 1
    acmod: int = 7
    lfeon: int = 1
    frame_bytes: int = 1536

    def c(self):
        return capi.DecodeDescC(self.flags, self.level, self.bias, self.dynrng, self.acmod, self.lfeon,
                                self.frame_bytes)


@dataclass
class EncodeDesc:
    """AC3_encode_init's arguments (src/ac3enc/ac3enc.h:6)."""
    sample_rate: int = 48000
    bit_rate: int = 384000
    channels: int = 6

    def c(self):
        return capi.EncodeDescC(self.sample_rate, self.bit_rate, self.channels)

    def frame_bytes(self):
        c = self.c()
        return capi.load_library().ac3mi_encode_frame_bytes(ctypes.byref(c))


def syncinfo(buf):
    """a52_syncinfo on host bytes -> (frame_bytes, flags, sample_rate, bit_rate); frame_bytes 0 = no frame."""
    lib = capi.load_library()
    b = (ctypes.c_uint8 * 8)(*bytes(buf[:8]))
    fl, sr, br = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    n = lib.ac3mi_syncinfo(b, ctypes.byref(fl), ctypes.byref(sr), ctypes.byref(br))
    return n, fl.value, sr.value, br.value


class Engine:
    def __init__(self, device=0):
        self.lib = capi.load_library()
        self.ctx = self.lib.ac3mi_create(device)
        if not self.ctx:
            raise capi.AC3MIError(self.lib.ac3mi_last_error(None).decode())
        self.device = device
        # tensors handed to in-flight launches: torch's caching allocator only orders reuse on
        # torch's own stream, so keep them alive until the engine stream has been drained
        self._keep = []

    def close(self):
        if self.ctx:
            self.lib.ac3mi_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise capi.AC3MIError("ac3mi error %d: %s" % (rc, self.lib.ac3mi_last_error(self.ctx).decode()))

    def _drain_torch(self, wait_torch):
        """The engine launches on its own (non-blocking) HIP streams.  Everything torch has queued for the buffers of a
        call - the producers of its inputs AND the fills of outputs allocated inside the call (torch.zeros) - must have
        finished before the engine touches them, so the drain comes after every allocation, right before the launch."""
        if wait_torch:
            import torch
            torch.cuda.synchronize(self.device)

    def sync(self):
        self._check(self.lib.ac3mi_sync(self.ctx))
        self._keep.clear()

    def memset(self, t, byte=0):
        """ac3mi_memset of a whole tensor on the engine's stream (asynchronous)."""
        self._check(self.lib.ac3mi_memset(ctypes.c_void_p(self.ctx), ctypes.c_void_p(t.data_ptr()), int(byte), ctypes.c_size_t(t.numel() * t.element_size())))

    def copy(self, dst, src):
        """ac3mi_memcpy_d2d of a whole tensor on the engine's stream (asynchronous)."""
        assert dst.numel() * dst.element_size() == src.numel() * src.element_size()
        self._check(self.lib.ac3mi_memcpy_d2d(ctypes.c_void_p(self.ctx), ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()),
                                              ctypes.c_size_t(dst.numel() * dst.element_size())))

    def timer_start(self):
        self._check(self.lib.ac3mi_timer_start(self.ctx))

    def timer_stop(self):
        ms = ctypes.c_float()
        self._check(self.lib.ac3mi_timer_stop(self.ctx, ctypes.byref(ms)))
        self._keep.clear()
        return ms.value

    def planes(self, desc):
        n_in, n_out = ctypes.c_int(), ctypes.c_int()
        c = desc.c()
        rc = self.lib.ac3mi_xform_planes(ctypes.byref(c), ctypes.byref(n_in), ctypes.byref(n_out))
        if rc != 0:
            raise capi.AC3MIError("output flags %d not reachable from acmod %d" % (desc.output, desc.acmod))
        return n_in.value, n_out.value

    def imdct_batch(self, desc, coeffs, delay, blksw=None, out=None, wait_torch=True):
        """coeffs [S][F][6][n_in][256] f32, delay [S][n_out][128] f32 (updated in place),
        blksw None or [S][F][6][nfchans] u8  ->  pcm [S][F][6][n_out][256] f32.

        The engine runs on its own HIP stream: with wait_torch the call first drains torch's
        stream (the inputs were produced there); call sync() before reading the result."""
        import torch
        n_in, n_out = self.planes(desc)
        S, F = coeffs.shape[0], coeffs.shape[1]
        assert coeffs.dtype == torch.float32 and coeffs.is_contiguous() and coeffs.is_cuda
        assert tuple(coeffs.shape) == (S, F, 6, n_in, 256), coeffs.shape
        assert delay.dtype == torch.float32 and delay.is_contiguous() and tuple(delay.shape) == (S, n_out, 128)
        if blksw is not None:
            assert blksw.dtype == torch.uint8 and blksw.is_contiguous() and blksw.shape[:3] == (S, F, 6)
        if out is None:
            out = torch.empty((S, F, 6, n_out, 256), dtype=torch.float32, device=coeffs.device)
        c = desc.c()
        self._drain_torch(wait_torch)
        self._check(self.lib.ac3mi_imdct_batch(self.ctx, ctypes.byref(c), coeffs.data_ptr(),
                                               blksw.data_ptr() if blksw is not None else None,
                                               delay.data_ptr(), out.data_ptr(), S, F))
        self._keep.append((coeffs, delay, blksw, out))
        return out

    def decode_planes(self, desc):
        n_out, fl = ctypes.c_int(), ctypes.c_int()
        c = desc.c()
        if self.lib.ac3mi_decode_planes(ctypes.byref(c), ctypes.byref(n_out), ctypes.byref(fl)) != 0:
            raise capi.AC3MIError("a52_frame would refuse flags %d for acmod %d" % (desc.flags, desc.acmod))
        return n_out.value, fl.value

    def decode_batch(self, desc, frames, delay, lfsr, out=None, status=None, taps=False, wait_torch=True, dynrng_in=None):
        """frames [S][F][stride] u8 (stride multiple of 4), delay [S][n_out][128] f32, lfsr [S] i16/u16
        (both updated in place) -> (pcm [S][F][6][n_out][256] f32, status [S][F] i32[, taps dict])."""
        import torch
        n_out, out_flags = self.decode_planes(desc)
        S, F, stride = frames.shape
        assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.is_cuda
        assert delay.dtype == torch.float32 and tuple(delay.shape) == (S, n_out, 128) and delay.is_contiguous()
        assert lfsr.dtype in (torch.int16, torch.uint16) and tuple(lfsr.shape) == (S,)
        dev = frames.device
        if out is None:
            out = torch.empty((S, F, 6, n_out, 256), dtype=torch.float32, device=dev)
        if status is None:
            status = torch.zeros((S, F), dtype=torch.int32, device=dev)
        nf = (2, 1, 2, 3, 3, 4, 4, 5)[desc.acmod]
        n_in = nf + (1 if desc.lfeon else 0)
        tp, tdict = None, None
        if taps:
            tdict = {
                "coef": torch.zeros((S, F, 6, n_in, 256), dtype=torch.float32, device=dev),
                "blksw": torch.zeros((S, F, 6, nf), dtype=torch.uint8, device=dev),
                "exp": torch.zeros((S, F, 6, 7, 256), dtype=torch.uint8, device=dev),
                "bap": torch.zeros((S, F, 6, 7, 256), dtype=torch.int8, device=dev),
            }
            tdict["dynrng"] = torch.full((S, F, 6, 2), float("nan"), dtype=torch.float32, device=dev)
            tp = capi.DecodeTapsC(tdict["coef"].data_ptr(), tdict["blksw"].data_ptr(), tdict["exp"].data_ptr(),
                                  tdict["bap"].data_ptr(), tdict["dynrng"].data_ptr(),
                                  dynrng_in.data_ptr() if dynrng_in is not None else None)
        c = desc.c()
        self._drain_torch(wait_torch or taps)
        self._check(self.lib.ac3mi_decode_batch(self.ctx, ctypes.byref(c), frames.data_ptr(), stride, S, F,
                                                delay.data_ptr(), lfsr.data_ptr(), out.data_ptr(),
                                                status.data_ptr(), ctypes.byref(tp) if tp else None))
        self._keep.append((frames, delay, lfsr, out, status, tdict))
        if taps:
            return out, status, tdict
        return out, status

    def decode_s16_batch(self, desc, frames, delay, lfsr, out=None, status=None, wait_torch=True):
        """ac3mi_decode_s16_batch: like decode_batch at level 1 / bias 384 with the reference's s16 converter folded into
        the transform -> (pcm [S][F][6][256][n_out] i16 in WAVE channel order, status [S][F] i32)."""
        import torch
        n_out, _ = self.decode_planes(desc)
        S, F, stride = frames.shape
        assert frames.dtype == torch.uint8 and frames.is_contiguous() and frames.is_cuda
        assert delay.dtype == torch.float32 and tuple(delay.shape) == (S, n_out, 128) and delay.is_contiguous()
        assert lfsr.dtype in (torch.int16, torch.uint16) and tuple(lfsr.shape) == (S,)
        dev = frames.device
        if out is None:
            out = torch.empty((S, F, 6, 256, n_out), dtype=torch.int16, device=dev)
        if status is None:
            status = torch.zeros((S, F), dtype=torch.int32, device=dev)
        c = desc.c()
        self._drain_torch(wait_torch)
        self._check(self.lib.ac3mi_decode_s16_batch(self.ctx, ctypes.byref(c), frames.data_ptr(), stride, S, F,
                                                    delay.data_ptr(), lfsr.data_ptr(), out.data_ptr(), status.data_ptr()))
        self._keep.append((frames, delay, lfsr, out, status))
        return out, status

    def probe_valu_rate(self):
        """10^9 VALU instructions/s one SIMD sustains under a chip-wide VALU load (ac3mi_probe_valu_rate)."""
        v = ctypes.c_double()
        self._check(self.lib.ac3mi_probe_valu_rate(ctypes.c_void_p(self.ctx), ctypes.byref(v)))
        return v.value

    def probe_copy_rate(self, nbytes=1 << 31):
        """GB/s (read + written) of a bare one-float4-per-lane copy of nbytes (ac3mi_probe_copy_rate)."""
        v = ctypes.c_double(0.0)
        self._check(self.lib.ac3mi_probe_copy_rate(ctypes.c_void_p(self.ctx), ctypes.c_size_t(int(nbytes)), ctypes.byref(v)))
        return v.value

    def probe_salu_rate(self):
        """10^9 scalar instructions/s one SIMD's share of the scalar unit sustains under a chip-wide load (ac3mi_probe_salu_rate)."""
        v = ctypes.c_double()
        self._check(self.lib.ac3mi_probe_salu_rate(ctypes.c_void_p(self.ctx), ctypes.byref(v)))
        return v.value

    def probe_mixed_rate(self, waves_per_simd=8):
        """(vector, scalar) 10^9 instructions/s per SIMD when every wavefront issues three vector instructions per scalar one
        (ac3mi_probe_mixed_rate)."""
        v, s = ctypes.c_double(), ctypes.c_double()
        self._check(self.lib.ac3mi_probe_mixed_rate(ctypes.c_void_p(self.ctx), int(waves_per_simd), ctypes.byref(v), ctypes.byref(s)))
        return v.value, s.value

    def workspace_bytes(self):
        """Device bytes the context's workspaces hold right now (ac3mi_workspace_bytes)."""
        self.lib.ac3mi_workspace_bytes.restype = ctypes.c_size_t
        return int(self.lib.ac3mi_workspace_bytes(ctypes.c_void_p(self.ctx)))

    def set_tile_frames(self, frames):
        """Workspace bound: batches above `frames` frames go through in tiles of whole streams (ac3mi_set_tile_frames)."""
        self._check(self.lib.ac3mi_set_tile_frames(ctypes.c_void_p(self.ctx), int(frames)))

    def set_decode_mode(self, mode):
        """0 = choose by batch shape, 1 = one wavefront per stream (one-kernel reference), 3 = one workgroup per stream with the
        transform fused in, 4 / 5 = parse kernel per stream / per frame + one wavefront per audio block, 6 = as 4 with the transform
        in the mantissa kernel for one-frame streams without a downmix (ac3mi_set_decode_mode)."""
        self._check(self.lib.ac3mi_set_decode_mode(ctypes.c_void_p(self.ctx), int(mode)))

    def set_encode_mode(self, mode):
        """0 = choose by batch size, 1 = one wavefront per frame packs, 2 = one wavefront per audio block packs
        (ac3mi_set_encode_mode)."""
        self._check(self.lib.ac3mi_set_encode_mode(ctypes.c_void_p(self.ctx), int(mode)))

    def set_mix_state(self, pending=None, flags=None):
        """liba52's overlap bookkeeping around frames with surround level 0 (ac3mi_set_mix_state): `pending` float32 shaped
        like the delay array, `flags` int32 [S][6], both zero for new streams and updated in place by the decode calls that
        follow; None, None returns to the plain linear mix.  The tensors must stay alive while set."""
        if (pending is None) != (flags is None):
            raise ValueError("set_mix_state: give both arrays or neither")
        if pending is not None:
            import torch
            assert pending.dtype == torch.float32 and flags.dtype == torch.int32 and pending.is_cuda and flags.is_cuda
            torch.cuda.current_stream().synchronize()        # their zero fill was queued on torch's stream
        self._mix_state = (pending, flags)
        self._check(self.lib.ac3mi_set_mix_state(ctypes.c_void_p(self.ctx),
                                                 ctypes.c_void_p(pending.data_ptr() if pending is not None else 0),
                                                 ctypes.c_void_p(flags.data_ptr() if flags is not None else 0)))

    def transcode_batch(self, dec, enc, frames, delay, lfsr, chmap, last, csnroffst, out=None, status=None, wait_torch=True):
        """frames [S][F][in_stride] u8 -> re-encoded frames [S][F][out_stride] u8 (+ status [S][F]); the state arrays
        are those of decode_batch (delay, lfsr) and encode_batch (last, csnroffst), all updated in place."""
        import torch
        S, F, in_stride = frames.shape
        fb = enc.frame_bytes()
        stride = (fb + 3) & ~3
        dev = frames.device
        if out is None:
            out = torch.zeros((S, F, stride), dtype=torch.uint8, device=dev)
        if status is None:
            status = torch.zeros((S, F), dtype=torch.int32, device=dev)
        cm = (ctypes.c_uint8 * 8)(*(list(chmap) + [0] * 8)[:8])
        dc, ec = dec.c(), enc.c()
        self._drain_torch(wait_torch)
        self._check(self.lib.ac3mi_transcode_batch(self.ctx, ctypes.byref(dc), ctypes.byref(ec), frames.data_ptr(), in_stride, S, F,
                                                   delay.data_ptr(), lfsr.data_ptr(), cm, last.data_ptr(), csnroffst.data_ptr(),
                                                   out.data_ptr(), stride, status.data_ptr()))
        self._keep.append((frames, delay, lfsr, last, csnroffst, out, status))
        return out, status

    def encode_batch(self, desc, pcm, chmap, last, csnroffst, out=None, taps=False, wait_torch=True):
        """pcm [S][F][1536][nch] s16, chmap = nch ints, last [S][nch][256] s16, csnroffst [S] i32 (both
        updated in place) -> frames [S][F][stride] u8 (stride = frame bytes rounded up to 4)[, taps dict]."""
        import torch
        fb = desc.frame_bytes()
        if fb <= 0:
            raise capi.AC3MIError("AC3_encode_init would reject %r" % (desc,))
        S, F, n, nch = pcm.shape
        assert n == 1536 and nch == desc.channels and pcm.dtype == torch.int16 and pcm.is_contiguous() and pcm.is_cuda
        assert last.dtype == torch.int16 and tuple(last.shape) == (S, nch, 256) and last.is_contiguous()
        assert csnroffst.dtype == torch.int32 and tuple(csnroffst.shape) == (S,)
        stride = (fb + 3) & ~3
        dev = pcm.device
        if out is None:
            out = torch.zeros((S, F, stride), dtype=torch.uint8, device=dev)
        else:
            assert out.dtype == torch.uint8 and out.is_contiguous() and out.shape[:2] == (S, F) and out.shape[2] >= stride and out.shape[2] % 4 == 0
            stride = out.shape[2]
        cm = (ctypes.c_uint8 * 8)(*(list(chmap) + [0] * 8)[:8])
        tp, tdict = None, None
        if taps:
            tdict = {
                "mdct": torch.zeros((S, F, 6, nch, 256), dtype=torch.int32, device=dev),
                "exponent": torch.zeros((S, F, 6, nch, 256), dtype=torch.uint8, device=dev),
                "exp_samples": torch.zeros((S, F, 6, nch), dtype=torch.int8, device=dev),
                "encoded_exp": torch.zeros((S, F, 6, nch, 256), dtype=torch.uint8, device=dev),
                "bap": torch.zeros((S, F, 6, nch, 256), dtype=torch.uint8, device=dev),
                "exp_strategy": torch.zeros((S, F, 6, nch), dtype=torch.uint8, device=dev),
                "snroffst": torch.zeros((S, F, 2), dtype=torch.int32, device=dev),
            }
            tp = capi.EncodeTapsC(*(tdict[k].data_ptr() for k in ("mdct", "exponent", "exp_samples", "encoded_exp",
                                                                   "bap", "exp_strategy", "snroffst")))
        c = desc.c()
        self._drain_torch(wait_torch or taps)
        self._check(self.lib.ac3mi_encode_batch(self.ctx, ctypes.byref(c), pcm.data_ptr(), cm, last.data_ptr(),
                                                csnroffst.data_ptr(), out.data_ptr(), stride, S, F,
                                                ctypes.byref(tp) if tp else None))
        self._keep.append((pcm, last, csnroffst, out, tdict))
        if taps:
            return out, tdict
        return out
