"""Minimal FLAC decoder (host side, pure Python) for the speech configuration of the sweep.

The reference reads `paper_plots/84-121123-0020.flac` with `soundfile.read` (target_snn_localization.py:149), i.e.
libsndfile; neither is available in the build image, so the driver-level I/O gets its own small decoder:
STREAMINFO + frames with CONSTANT / VERBATIM / FIXED / LPC subframes, Rice-coded residuals (both parameter
widths, escape partitions), wasted bits, all four channel assignments, 4..32 bits per sample.  The decoded PCM is
verified against the MD5 signature stored in STREAMINFO.  `read(path)` mirrors `soundfile.read`: float64 in [-1, 1)
and the sample rate.
"""
import hashlib

import numpy as np


class FlacError(ValueError):
    pass


class _Bits:
    __slots__ = ("data", "pos", "n")

    def __init__(self, data, byte_pos=0):
        self.data = data
        self.pos = byte_pos * 8
        self.n = len(data) * 8

    def read(self, nbits):
        if nbits == 0:
            return 0
        pos = self.pos
        end = pos + nbits
        if end > self.n:
            raise FlacError("unexpected end of FLAC stream")
        b0, b1 = pos >> 3, (end + 7) >> 3
        val = int.from_bytes(self.data[b0:b1], "big")
        val >>= (b1 << 3) - end
        self.pos = end
        return val & ((1 << nbits) - 1)

    def read_signed(self, nbits):
        v = self.read(nbits)
        return v - (1 << nbits) if v >> (nbits - 1) else v

    def unary(self):
        """number of 0 bits before the next 1 bit (consumes the 1)."""
        data = self.data
        pos = self.pos
        count = 0
        # finish the current byte
        while True:
            byte_i = pos >> 3
            if byte_i >= len(data):
                raise FlacError("unexpected end of FLAC stream")
            rem = 8 - (pos & 7)
            cur = data[byte_i] & ((1 << rem) - 1)
            if cur:
                lead = rem - cur.bit_length()
                self.pos = pos + lead + 1
                return count + lead
            count += rem
            pos += rem

    def align(self):
        self.pos = (self.pos + 7) & ~7


_FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


def _residual(br, blocksize, order, out):
    method = br.read(2)
    if method > 1:
        raise FlacError("reserved residual coding method")
    pbits = 4 if method == 0 else 5
    esc = (1 << pbits) - 1
    porder = br.read(4)
    nparts = 1 << porder
    idx = order
    for p in range(nparts):
        count = (blocksize >> porder) - (order if p == 0 else 0)
        k = br.read(pbits)
        if k == esc:
            nb = br.read(5)
            for _ in range(count):
                out[idx] = br.read_signed(nb) if nb else 0
                idx += 1
        else:
            unary = br.unary
            read = br.read
            for _ in range(count):
                q = unary()
                u = (q << k) | (read(k) if k else 0)
                out[idx] = (u >> 1) ^ -(u & 1)
                idx += 1


def _subframe(br, blocksize, bps):
    if br.read(1):
        raise FlacError("bad subframe padding bit")
    kind = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.unary() + 1
        bps -= wasted
    out = [0] * blocksize
    if kind == 0:  # CONSTANT
        out = [br.read_signed(bps)] * blocksize
    elif kind == 1:  # VERBATIM
        out = [br.read_signed(bps) for _ in range(blocksize)]
    elif 8 <= kind <= 12:  # FIXED
        order = kind - 8
        for i in range(order):
            out[i] = br.read_signed(bps)
        _residual(br, blocksize, order, out)
        coefs = _FIXED[order]
        for i in range(order, blocksize):
            acc = out[i]
            for j, c in enumerate(coefs):
                acc += c * out[i - 1 - j]
            out[i] = acc
    elif kind >= 32:  # LPC
        order = (kind & 31) + 1
        for i in range(order):
            out[i] = br.read_signed(bps)
        prec = br.read(4) + 1
        if prec == 16:
            raise FlacError("invalid LPC precision")
        shift = br.read_signed(5)
        if shift < 0:
            raise FlacError("negative LPC shift")
        coefs = [br.read_signed(prec) for _ in range(order)]
        _residual(br, blocksize, order, out)
        for i in range(order, blocksize):
            acc = 0
            for j in range(order):
                acc += coefs[j] * out[i - 1 - j]
            out[i] += acc >> shift
    else:
        raise FlacError("reserved subframe type")
    if wasted:
        out = [v << wasted for v in out]
    return out


_BLOCKSIZES = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608, 8: 256, 9: 512, 10: 1024, 11: 2048, 12: 4096, 13: 8192, 14: 16384, 15: 32768}
_BPS = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}


def decode(data):
    """bytes of a .flac file -> (pcm int array [n, channels], sample_rate, bits_per_sample)."""
    if data[:4] != b"fLaC":
        raise FlacError("not a FLAC stream")
    pos = 4
    info = None
    while True:
        header = data[pos]
        length = int.from_bytes(data[pos + 1 : pos + 4], "big")
        if header & 0x7F == 0:
            info = data[pos + 4 : pos + 4 + length]
        pos += 4 + length
        if header & 0x80:
            break
    if info is None:
        raise FlacError("missing STREAMINFO")
    bi = _Bits(info)
    bi.read(16), bi.read(16), bi.read(24), bi.read(24)
    rate = bi.read(20)
    channels = bi.read(3) + 1
    bps = bi.read(5) + 1
    total = bi.read(36)
    md5 = info[18:34]

    chans = [[] for _ in range(channels)]
    br = _Bits(data, pos)
    while br.pos < br.n and (total == 0 or len(chans[0]) < total):
        if br.read(14) != 0x3FFE:
            raise FlacError("lost frame sync")
        br.read(1)
        br.read(1)
        bs_code = br.read(4)
        sr_code = br.read(4)
        ch_code = br.read(4)
        bps_code = br.read(3)
        br.read(1)
        first = br.read(8)  # UTF-8 coded frame / sample number
        extra = 0
        while first & 0x80 and first & (0x40 >> extra):
            extra += 1
        if first & 0x80:
            for _ in range(extra):
                br.read(8)
        if bs_code == 6:
            blocksize = br.read(8) + 1
        elif bs_code == 7:
            blocksize = br.read(16) + 1
        else:
            blocksize = _BLOCKSIZES[bs_code]
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        br.read(8)  # CRC-8
        fbps = _BPS.get(bps_code, bps)
        if ch_code < 8:
            subs = [_subframe(br, blocksize, fbps) for _ in range(ch_code + 1)]
        elif ch_code == 8:  # left / side
            left = _subframe(br, blocksize, fbps)
            side = _subframe(br, blocksize, fbps + 1)
            subs = [left, [a - s for a, s in zip(left, side)]]
        elif ch_code == 9:  # side / right
            side = _subframe(br, blocksize, fbps + 1)
            right = _subframe(br, blocksize, fbps)
            subs = [[s + r for s, r in zip(side, right)], right]
        elif ch_code == 10:  # mid / side
            mid = _subframe(br, blocksize, fbps)
            side = _subframe(br, blocksize, fbps + 1)
            lefts, rights = [], []
            for m_, s_ in zip(mid, side):
                m2 = (m_ << 1) | (s_ & 1)
                lefts.append((m2 + s_) >> 1)
                rights.append((m2 - s_) >> 1)
            subs = [lefts, rights]
        else:
            raise FlacError("reserved channel assignment")
        br.align()
        br.read(16)  # CRC-16
        for c in range(channels):
            chans[c].extend(subs[c])
    pcm = np.asarray(chans, dtype=np.int64).T
    if total:
        pcm = pcm[:total]
    if md5 != bytes(16):
        width = (bps + 7) // 8
        raw = b"".join(int(v).to_bytes(width, "little", signed=True) for v in pcm.ravel())
        if hashlib.md5(raw).digest() != md5:
            raise FlacError("decoded audio does not match the STREAMINFO MD5 signature")
    return pcm, rate, bps


def read(path):
    """Like soundfile.read(path): (float64 samples in [-1, 1), shape [n] for mono else [n, ch]; sample rate)."""
    with open(path, "rb") as f:
        pcm, rate, bps = decode(f.read())
    sig = pcm.astype(np.float64) / float(1 << (bps - 1))
    return (sig[:, 0] if sig.shape[1] == 1 else sig), rate
