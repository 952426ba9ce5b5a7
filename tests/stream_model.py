"""Byte accounting of the reference's two stream_convert functions, restated in Python from
src/AC3ACM.cpp:1430-1628 (AC-3 -> PCM) and :1665-1798 (PCM -> AC-3): which source bytes a call consumes and
how many destination bytes it produces, for any chunking.  Test infrastructure (checks
ac3mi_stream_convert); it knows nothing about audio, only frame sizes."""

_KBPS = (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 448, 512, 576, 640)


def syncinfo_size(b):
    """frame size in bytes as a52_syncinfo reports it (a52dec-0.7.5-cvs/liba52/parse.c:86-129), 0 = no frame"""
    if b[0] != 0x0B or b[1] != 0x77:
        return 0
    if b[5] >= 0x60:
        return 0
    frmsizecod = b[4] & 63
    if frmsizecod >= 38:
        return 0
    rate = _KBPS[frmsizecod >> 1]
    fscod = b[4] & 0xC0
    if fscod == 0x00:
        return 4 * rate
    if fscod == 0x40:
        return 2 * (320 * rate // 147 + (frmsizecod & 1))
    if fscod == 0x80:
        return 6 * rate
    return 0


class DecodeModel:
    def __init__(self, src_channels, dst_channels):
        self.buf = bytearray(4096 + 64)
        self.bufptr = self.bufend = 0
        self.blocks = 0
        self.blk = 512 * dst_channels

    def convert(self, src, dcap, start):
        src_used = dst_used = 0
        src_len, src_left, dst_left, sp = len(src), len(src), dcap, 0
        if start:
            self.bufptr = self.bufend = 0
            self.blocks = 0
        elif self.blocks > 0:
            while True:
                dst_left -= self.blk
                if dst_left < 0:
                    return src_used, dst_used
                dst_used += self.blk
                self.blocks -= 1
                if not self.blocks:
                    break
        while True:
            fs = 128 - self.bufend
            sr = self.bufend - self.bufptr
            if sr >= 8:
                while True:
                    fs = syncinfo_size(self.buf[self.bufptr:self.bufptr + 8])
                    if fs:
                        if sr < fs:
                            fs -= sr
                            break
                        self.bufptr = self.bufend = 0
                        few = 1 if (src_len * 2 - src_used) < fs else 0
                        self.blocks = 6
                        while self.blocks > few:
                            dst_left -= self.blk
                            if dst_left < 0:
                                return src_used, dst_used
                            dst_used += self.blk
                            self.blocks -= 1
                        fs = 128
                        break
                    self.bufptr += 1
                    sr -= 1
                    if sr < 8:
                        self.buf[0:8] = self.buf[self.bufptr:self.bufptr + 8]
                        self.bufptr, self.bufend = 0, 8
                        fs = 120
                        break
            if src_left <= 0:
                break
            fs = min(fs, src_left)
            self.buf[self.bufend:self.bufend + fs] = src[sp:sp + fs]
            self.bufend += fs
            src_used += fs
            src_left -= fs
            sp += fs
        return src_used, dst_used


class EncodeModel:
    def __init__(self, channels, frame_bytes):
        self.needed = 1536 * channels * 2
        self.frame_bytes = frame_bytes
        self.fill = 0           # msd->bufptr - msd->buf
        self.left = 0           # msd->blocks: encoded bytes not yet handed out

    def convert(self, src_len, dcap, start):
        src_used = dst_used = 0
        src_left, dst_left = src_len, dcap
        if start:
            self.fill = 0
            self.left = 0
        elif self.left > 0:
            tc = min(self.left, dst_left)
            if tc > 0:
                dst_used += tc
                self.left -= tc
                dst_left -= tc
            if dst_left <= 0:
                return src_used, dst_used
        while src_left > 0:
            fs = self.fill
            if fs < self.needed:
                tc = min(self.needed - fs, src_left)
                src_used += tc
                self.fill += tc
                src_left -= tc
                fs += tc
            if fs >= self.needed:
                tc = self.frame_bytes
                self.fill = 0
                self.left = tc
                tc = min(tc, dst_left)
                if tc > 0:
                    dst_used += tc
                    self.left -= tc
                    dst_left -= tc
                if dst_left <= 0:
                    break
        return src_used, dst_used
