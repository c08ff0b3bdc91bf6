"""Minimal mp4 (ISO-BMFF + H.264) writer / reader in numpy -- SURVEY.md 8(f) F2, the video I/O either side of the hot path.

The reference writes its results with ``diffusers.utils.export_to_video`` (``i2vgen-xl/run_group_pnp_edit.py:178``; OpenCV /
imageio + ffmpeg underneath) and reads source clips with ``torchvision.io.read_video`` (``i2vgen-xl/utils.py:43``).  None of
those, and no codec library, exists in this image, so:

* ``write_mp4`` produces a standard ``.mp4`` (``avc1`` track, Baseline profile, every frame an IDR picture whose macroblocks
  are all ``I_PCM`` -- raw 4:2:0 samples, no transform / entropy coding).  Any H.264 decoder (ffmpeg, browsers, QuickTime)
  plays it; it is lossless in YUV and large (1.5 bytes per pixel per frame: 6.3 MB for 16 x 512 x 512).
* ``read_mp4`` demuxes any mp4 and decodes the pictures **if they are I_PCM** (i.e. files written by ``write_mp4``, which is
  what chains stage outputs back into stage inputs here).  Real camera / encoder output uses intra / inter prediction + CAVLC
  or CABAC, which needs a full decoder: that raises ``Mp4Unsupported`` naming the first unsupported feature, and the runners
  tell the user to supply a ``%05d.png`` frame directory instead.

Colour: BT.601 limited range (what players assume for untagged SD-sized streams), chroma averaged over 2 x 2.
"""
from __future__ import annotations

import re
import struct
from typing import List, Tuple

import numpy as np
from PIL import Image


class Mp4Unsupported(RuntimeError):
    pass


# ------------------------------------------------------------------------------------------------ bit-level helpers
class _BitWriter:
    def __init__(self):
        self.bits: List[int] = []

    def u(self, n: int, v: int):
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    def ue(self, v: int):
        v += 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)

    def se(self, v: int):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def align_zero(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def trailing(self):  # rbsp_trailing_bits
        self.bits.append(1)
        self.align_zero()

    def bytes(self) -> bytes:
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, dtype=np.uint8)).tobytes()


class _BitReader:
    def __init__(self, data: bytes):
        self.bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))
        self.pos = 0

    def u(self, n: int) -> int:
        if self.pos + n > len(self.bits):
            raise Mp4Unsupported("truncated H.264 header")
        v = 0
        for b in self.bits[self.pos:self.pos + n]:
            v = (v << 1) | int(b)
        self.pos += n
        return v

    def ue(self) -> int:
        z = 0
        while self.u(1) == 0:
            z += 1
            if z > 32:
                raise Mp4Unsupported("bad exp-Golomb code")
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self) -> int:
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)

    def align(self):
        self.pos = (self.pos + 7) & ~7


_EPB = re.compile(rb"\x00\x00(?=[\x00-\x03])")


def _escape(rbsp: bytes) -> bytes:
    """emulation prevention: 00 00 0x -> 00 00 03 0x (H.264 7.4.1)"""
    return _EPB.sub(b"\x00\x00\x03", rbsp)


def _unescape(nal: bytes) -> bytes:
    return nal.replace(b"\x00\x00\x03", b"\x00\x00")


# ------------------------------------------------------------------------------------------------ colour
def _rgb_to_yuv420(rgb: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """[H, W, 3] uint8 (H, W even) -> Y [H, W], Cb / Cr [H/2, W/2] uint8, BT.601 limited range."""
    f = rgb.astype(np.float32)
    r, g, b = f[..., 0], f[..., 1], f[..., 2]
    y = 16.0 + (65.481 * r + 128.553 * g + 24.966 * b) / 255.0
    cb = 128.0 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255.0
    cr = 128.0 + (112.0 * r - 93.786 * g - 18.214 * b) / 255.0
    h, w = y.shape
    sub = lambda c: c.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))
    q = lambda c: np.clip(np.rint(c), 0, 255).astype(np.uint8)
    return q(y), q(sub(cb)), q(sub(cr))


def _yuv420_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    yf = (y.astype(np.float32) - 16.0) * (255.0 / 219.0)
    up = lambda c: np.repeat(np.repeat(c.astype(np.float32) - 128.0, 2, axis=0), 2, axis=1)[:y.shape[0], :y.shape[1]]
    cbf, crf = up(cb) * (255.0 / 224.0), up(cr) * (255.0 / 224.0)
    r = yf + 1.402 * crf
    g = yf - 0.344136 * cbf - 0.714136 * crf
    b = yf + 1.772 * cbf
    return np.clip(np.rint(np.stack([r, g, b], -1)), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ H.264 (Baseline, I_PCM)
def _sps(width: int, height: int) -> bytes:
    mbw, mbh = (width + 15) // 16, (height + 15) // 16
    w = _BitWriter()
    w.u(8, 66)           # profile_idc: Baseline
    w.u(8, 0xC0)         # constraint_set0/1
    w.u(8, 51)           # level_idc 5.1 (raw samples exceed every level's bit rate anyway)
    w.ue(0)              # seq_parameter_set_id
    w.ue(0)              # log2_max_frame_num_minus4
    w.ue(2)              # pic_order_cnt_type 2: output order == decoding order
    w.ue(1)              # max_num_ref_frames
    w.u(1, 0)            # gaps_in_frame_num_value_allowed_flag
    w.ue(mbw - 1)
    w.ue(mbh - 1)
    w.u(1, 1)            # frame_mbs_only_flag
    w.u(1, 1)            # direct_8x8_inference_flag
    crop_r, crop_b = (mbw * 16 - width) // 2, (mbh * 16 - height) // 2
    if crop_r or crop_b:
        w.u(1, 1)
        w.ue(0); w.ue(crop_r); w.ue(0); w.ue(crop_b)
    else:
        w.u(1, 0)
    w.u(1, 0)            # vui_parameters_present_flag (frame rate lives in the container)
    w.trailing()
    return b"\x67" + _escape(w.bytes())


def _pps() -> bytes:
    w = _BitWriter()
    w.ue(0); w.ue(0)     # pic_parameter_set_id, seq_parameter_set_id
    w.u(1, 0)            # entropy_coding_mode_flag: CAVLC
    w.u(1, 0)            # bottom_field_pic_order_in_frame_present_flag
    w.ue(0)              # num_slice_groups_minus1
    w.ue(0); w.ue(0)     # num_ref_idx_l0/l1_default_active_minus1
    w.u(1, 0); w.u(2, 0)  # weighted_pred_flag, weighted_bipred_idc
    w.se(0); w.se(0); w.se(0)  # pic_init_qp/qs_minus26, chroma_qp_index_offset
    w.u(1, 1)            # deblocking_filter_control_present_flag (slices switch the filter off)
    w.u(1, 0); w.u(1, 0)  # constrained_intra_pred_flag, redundant_pic_cnt_present_flag
    w.trailing()
    return b"\x68" + _escape(w.bytes())


def _idr_slice(y: np.ndarray, cb: np.ndarray, cr: np.ndarray, idr_pic_id: int) -> bytes:
    """One IDR picture = one slice, every macroblock I_PCM.  y / cb / cr are padded to whole macroblocks."""
    mbh, mbw = y.shape[0] // 16, y.shape[1] // 16
    ymb = y.reshape(mbh, 16, mbw, 16).transpose(0, 2, 1, 3).reshape(mbh * mbw, 256)
    cbm = cb.reshape(mbh, 8, mbw, 8).transpose(0, 2, 1, 3).reshape(mbh * mbw, 64)
    crm = cr.reshape(mbh, 8, mbw, 8).transpose(0, 2, 1, 3).reshape(mbh * mbw, 64)
    w = _BitWriter()
    w.ue(0)              # first_mb_in_slice
    w.ue(7)              # slice_type: I (all slices of the picture)
    w.ue(0)              # pic_parameter_set_id
    w.u(4, 0)            # frame_num (IDR)
    w.ue(idr_pic_id)
    w.u(1, 0); w.u(1, 0)  # no_output_of_prior_pics_flag, long_term_reference_flag
    w.se(0)              # slice_qp_delta
    w.ue(1)              # disable_deblocking_filter_idc = 1
    w.ue(25)             # mb_type of macroblock 0: I_PCM
    w.align_zero()       # pcm_alignment_zero_bit
    head = w.bytes()
    # every further macroblock starts byte aligned: ue(25) = 0000 11010, + 7 alignment zeros = 0x0D 0x00
    body = np.empty((mbh * mbw, 2 + 384), dtype=np.uint8)
    body[:, 0], body[:, 1] = 0x0D, 0x00
    body[:, 2:258], body[:, 258:322], body[:, 322:386] = ymb, cbm, crm
    rbsp = head + body.reshape(-1)[2:].tobytes() + b"\x80"  # (macroblock 0's mb_type sits in `head`); rbsp_slice_trailing_bits
    return b"\x65" + _escape(rbsp)


# ------------------------------------------------------------------------------------------------ ISO-BMFF
def _box(kind: bytes, *payload: bytes) -> bytes:
    body = b"".join(payload)
    return struct.pack(">I4s", 8 + len(body), kind) + body


def _full(kind: bytes, version: int, flags: int, *payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags), *payload)


_MATRIX = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def write_mp4(frames: List[Image.Image], path: str, fps: float = 8) -> str:
    """``export_to_video(frames, path, fps)`` (``i2vgen-xl/run_group_pnp_edit.py:178``): H.264 I_PCM in an mp4."""
    if not frames:
        raise ValueError("write_mp4: no frames")
    width, height = frames[0].size
    if width % 2 or height % 2:
        raise ValueError(f"write_mp4: frame size must be even for 4:2:0 ({width} x {height})")
    pw, ph = (width + 15) // 16 * 16, (height + 15) // 16 * 16
    sps, pps = _sps(width, height), _pps()
    samples = []
    for i, fr in enumerate(frames):
        if fr.size != (width, height):
            raise ValueError("write_mp4: frames differ in size")
        rgb = np.asarray(fr.convert("RGB"), dtype=np.uint8)
        rgb = np.pad(rgb, ((0, ph - height), (0, pw - width), (0, 0)), mode="edge")
        nal = _idr_slice(*_rgb_to_yuv420(rgb), idr_pic_id=i & 1)
        samples.append(struct.pack(">I", len(nal)) + nal)
    n = len(samples)
    timescale, delta = int(round(fps * 1000)), 1000
    dur_media = n * delta
    dur_movie = int(round(n * 1000 / fps))  # movie timescale 1000
    avcc = _box(b"avcC", bytes([1, sps[1], sps[2], sps[3], 0xFF, 0xE1]), struct.pack(">H", len(sps)), sps, b"\x01",
                struct.pack(">H", len(pps)), pps)
    avc1 = _box(b"avc1", b"\x00" * 6, struct.pack(">H", 1), b"\x00" * 16, struct.pack(">HH", width, height),
                struct.pack(">II", 0x00480000, 0x00480000), b"\x00" * 4, struct.pack(">H", 1), b"\x00" * 32,
                struct.pack(">Hh", 0x0018, -1), avcc)
    stbl_head = [_full(b"stsd", 0, 0, struct.pack(">I", 1), avc1),
                 _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta)),
                 _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1)),
                 _full(b"stsz", 0, 0, struct.pack(">II", 0, n), b"".join(struct.pack(">I", len(s)) for s in samples))]

    def moov(chunk_offset: int) -> bytes:
        stbl = _box(b"stbl", *stbl_head, _full(b"stco", 0, 0, struct.pack(">II", 1, chunk_offset)))
        minf = _box(b"minf", _full(b"vmhd", 0, 1, b"\x00" * 8),
                    _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1), _full(b"url ", 0, 1))), stbl)
        mdia = _box(b"mdia", _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, dur_media, 0x55C4, 0)),
                    _full(b"hdlr", 0, 0, b"\x00" * 4, b"vide", b"\x00" * 12, b"VideoHandler\x00"), minf)
        tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, dur_movie), b"\x00" * 8, struct.pack(">hhhH", 0, 0, 0, 0),
                     _MATRIX, struct.pack(">II", width << 16, height << 16))
        mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, 1000, dur_movie), struct.pack(">IH", 0x10000, 0x0100),
                     b"\x00" * 10, _MATRIX, b"\x00" * 24, struct.pack(">I", 2))
        return _box(b"moov", mvhd, _box(b"trak", tkhd, mdia))

    ftyp = _box(b"ftyp", b"isom", struct.pack(">I", 0x200), b"isomiso2avc1mp41")
    off = len(ftyp) + len(moov(0)) + 8
    data = b"".join(samples)
    if off + len(data) >= 1 << 32:
        raise ValueError("write_mp4: file would exceed 4 GiB")
    with open(path, "wb") as f:
        f.write(ftyp + moov(off) + struct.pack(">I4s", 8 + len(data), b"mdat") + data)
    return path


def _boxes(buf: bytes, start: int, end: int):
    pos = start
    while pos + 8 <= end:
        size, kind = struct.unpack(">I4s", buf[pos:pos + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", buf[pos + 8:pos + 16])[0]
            hdr = 16
        elif size == 0:
            size = end - pos
        if size < hdr or pos + size > end:
            raise Mp4Unsupported("corrupt mp4 box structure")
        yield kind, pos + hdr, pos + size
        pos += size


def _find(buf: bytes, start: int, end: int, *path: bytes):
    for kind, a, b in _boxes(buf, start, end):
        if kind == path[0]:
            return (a, b) if len(path) == 1 else _find(buf, a, b, *path[1:])
    return None


def _parse_sps(nal: bytes) -> dict:
    r = _BitReader(_unescape(nal[1:]))
    profile = r.u(8)
    r.u(8); r.u(8)
    r.ue()
    if profile in (100, 110, 122, 244, 44, 83, 86, 118, 128, 138, 139, 134, 135):
        chroma = r.ue()
        if chroma == 3:
            r.u(1)
        if chroma != 1 or r.ue() or r.ue():
            raise Mp4Unsupported("H.264 stream is not 8-bit 4:2:0")
        r.u(1)
        if r.u(1):
            raise Mp4Unsupported("H.264 scaling matrices")
    log2_fn = r.ue() + 4
    poc_type = r.ue()
    log2_poc = 0
    if poc_type == 0:
        log2_poc = r.ue() + 4
    elif poc_type == 1:
        raise Mp4Unsupported("H.264 pic_order_cnt_type 1")
    r.ue(); r.u(1)
    mbw, mbh = r.ue() + 1, r.ue() + 1
    if not r.u(1):
        raise Mp4Unsupported("interlaced H.264")
    r.u(1)
    cl = cr = ct = cb = 0
    if r.u(1):
        cl, cr, ct, cb = r.ue(), r.ue(), r.ue(), r.ue()
    return dict(mbw=mbw, mbh=mbh, log2_fn=log2_fn, poc_type=poc_type, log2_poc=log2_poc, crop=(2 * cl, 2 * cr, 2 * ct, 2 * cb))


def _parse_pps(nal: bytes) -> dict:
    r = _BitReader(_unescape(nal[1:]))
    r.ue(); r.ue()
    cabac = r.u(1)
    bottom = r.u(1)
    if r.ue():
        raise Mp4Unsupported("H.264 slice groups")
    r.ue(); r.ue(); r.u(1); r.u(2); r.se(); r.se(); r.se()
    deblock = r.u(1)
    r.u(1)
    redundant = r.u(1)
    return dict(cabac=cabac, bottom=bottom, deblock=deblock, redundant=redundant)


def _decode_ipcm_picture(nal: bytes, sps: dict, pps: dict) -> np.ndarray:
    if pps["cabac"]:
        raise Mp4Unsupported("CABAC-coded H.264 (needs a full decoder)")
    rbsp = _unescape(nal[1:])
    r = _BitReader(rbsp[:64])
    if r.ue() != 0:
        raise Mp4Unsupported("multi-slice H.264 pictures")
    if r.ue() % 5 != 2:
        raise Mp4Unsupported("inter-predicted H.264 slices (needs a full decoder)")
    r.ue()
    r.u(sps["log2_fn"])
    idr = (nal[0] & 31) == 5
    if idr:
        r.ue()
    if sps["poc_type"] == 0:
        r.u(sps["log2_poc"])
        if pps["bottom"]:
            r.se()
    if pps["redundant"]:
        r.ue()
    if nal[0] >> 5 & 3:  # dec_ref_pic_marking
        if idr:
            r.u(2)
        elif r.u(1):
            raise Mp4Unsupported("H.264 memory management control operations")
    r.se()
    if pps["deblock"]:
        if r.ue() != 1:
            r.se(); r.se()
    if r.ue() != 25:
        raise Mp4Unsupported("intra-predicted H.264 macroblocks (only I_PCM pictures, as written by write_mp4, can be decoded "
                             "without a codec library)")
    r.align()
    nmb = sps["mbw"] * sps["mbh"]
    first = r.pos // 8
    need = first + 384 + (nmb - 1) * 386
    if len(rbsp) < need:
        raise Mp4Unsupported("H.264 picture is not all I_PCM")
    body = np.frombuffer(rbsp, dtype=np.uint8, count=(nmb - 1) * 386, offset=first + 384).reshape(nmb - 1, 386) if nmb > 1 \
        else np.empty((0, 386), dtype=np.uint8)
    if nmb > 1 and not ((body[:, 0] == 0x0D) & (body[:, 1] == 0x00)).all():
        raise Mp4Unsupported("H.264 picture is not all I_PCM")
    mb = np.concatenate([np.frombuffer(rbsp, dtype=np.uint8, count=384, offset=first)[None], body[:, 2:]], 0)
    mbh, mbw = sps["mbh"], sps["mbw"]
    y = mb[:, :256].reshape(mbh, mbw, 16, 16).transpose(0, 2, 1, 3).reshape(mbh * 16, mbw * 16)
    cb = mb[:, 256:320].reshape(mbh, mbw, 8, 8).transpose(0, 2, 1, 3).reshape(mbh * 8, mbw * 8)
    cr = mb[:, 320:384].reshape(mbh, mbw, 8, 8).transpose(0, 2, 1, 3).reshape(mbh * 8, mbw * 8)
    rgb = _yuv420_to_rgb(y, cb, cr)
    cl, crr, ct, cbt = sps["crop"]
    return rgb[ct:rgb.shape[0] - cbt, cl:rgb.shape[1] - crr]


def read_mp4(path: str) -> Tuple[List[Image.Image], float]:
    """Frames (RGB PIL images) and frame rate of an mp4 whose H.264 pictures are all I_PCM; ``Mp4Unsupported`` otherwise."""
    with open(path, "rb") as f:
        buf = f.read()
    moov = _find(buf, 0, len(buf), b"moov")
    if moov is None:
        raise Mp4Unsupported("no moov box (not an mp4 file?)")
    for kind, a, b in _boxes(buf, *moov):
        if kind != b"trak":
            continue
        hd = _find(buf, a, b, b"mdia", b"hdlr")
        if hd is None or buf[hd[0] + 8:hd[0] + 12] != b"vide":
            continue
        mdhd = _find(buf, a, b, b"mdia", b"mdhd")
        ver = buf[mdhd[0]]
        timescale = struct.unpack(">I", buf[mdhd[0] + (20 if ver else 12):mdhd[0] + (24 if ver else 16)])[0]
        stbl = _find(buf, a, b, b"mdia", b"minf", b"stbl")
        stsd = _find(buf, *stbl, b"stsd")
        entry = next(_boxes(buf, stsd[0] + 8, stsd[1]))
        if entry[0] not in (b"avc1", b"avc3"):
            raise Mp4Unsupported(f"video codec {entry[0].decode(errors='replace')!r}: only H.264 (avc1) I_PCM streams can be "
                                 "decoded without a codec library")
        avcc = _find(buf, entry[1] + 78, entry[2], b"avcC")
        c = buf[avcc[0]:avcc[1]]
        nlen = (c[4] & 3) + 1
        pos, sps, pps = 6, None, None
        for _ in range(c[5] & 31):
            ln = struct.unpack(">H", c[pos:pos + 2])[0]
            sps = sps or c[pos + 2:pos + 2 + ln]
            pos += 2 + ln
        npps = c[pos]
        pos += 1
        for _ in range(npps):
            ln = struct.unpack(">H", c[pos:pos + 2])[0]
            pps = pps or c[pos + 2:pos + 2 + ln]
            pos += 2 + ln
        if sps is None or pps is None:
            raise Mp4Unsupported("avcC without SPS / PPS")
        sps_d, pps_d = _parse_sps(sps), _parse_pps(pps)
        # sample table -> (offset, size) of every sample
        stsz = _find(buf, *stbl, b"stsz")
        fixed, n = struct.unpack(">II", buf[stsz[0] + 4:stsz[0] + 12])
        sizes = [fixed] * n if fixed else list(struct.unpack(f">{n}I", buf[stsz[0] + 12:stsz[0] + 12 + 4 * n]))
        co = _find(buf, *stbl, b"stco")
        if co is not None:
            nc = struct.unpack(">I", buf[co[0] + 4:co[0] + 8])[0]
            chunks = list(struct.unpack(f">{nc}I", buf[co[0] + 8:co[0] + 8 + 4 * nc]))
        else:
            co = _find(buf, *stbl, b"co64")
            nc = struct.unpack(">I", buf[co[0] + 4:co[0] + 8])[0]
            chunks = list(struct.unpack(f">{nc}Q", buf[co[0] + 8:co[0] + 8 + 8 * nc]))
        stsc = _find(buf, *stbl, b"stsc")
        ne = struct.unpack(">I", buf[stsc[0] + 4:stsc[0] + 8])[0]
        runs = [struct.unpack(">III", buf[stsc[0] + 8 + 12 * i:stsc[0] + 20 + 12 * i]) for i in range(ne)]
        offsets, si = [], 0
        for ci in range(nc):
            spc = next(r[1] for r in reversed(runs) if r[0] <= ci + 1)
            off = chunks[ci]
            for _ in range(spc):
                if si >= n:
                    break
                offsets.append(off)
                off += sizes[si]
                si += 1
        stts = _find(buf, *stbl, b"stts")
        _count, delta = struct.unpack(">II", buf[stts[0] + 8:stts[0] + 16])  # (first run of the time-to-sample table)
        fps = timescale / delta if delta else 0.0
        frames = []
        for off, size in zip(offsets, sizes):
            pos, pic = off, None
            while pos + nlen <= off + size:
                ln = int.from_bytes(buf[pos:pos + nlen], "big")
                nal = buf[pos + nlen:pos + nlen + ln]
                pos += nlen + ln
                t = nal[0] & 31 if nal else 0
                if t == 7:
                    sps_d = _parse_sps(nal)
                elif t == 8:
                    pps_d = _parse_pps(nal)
                elif t in (1, 5):
                    pic = _decode_ipcm_picture(nal, sps_d, pps_d)
            if pic is not None:
                frames.append(Image.fromarray(pic, "RGB"))
        return frames, fps
    raise Mp4Unsupported("no video track")
