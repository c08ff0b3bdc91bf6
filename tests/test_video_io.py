"""F2 video I/O (``anyv2v_amd/mp4.py``): the mp4 the runners write in place of the reference's ``export_to_video``
(``i2vgen-xl/run_group_pnp_edit.py:178``) and read back in place of ``read_video`` (``i2vgen-xl/utils.py:43``)."""
import os
import struct

import numpy as np
import pytest
from PIL import Image

from anyv2v_amd import mp4, utils


def _smooth(w, h, k):
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 120 * np.sin(x / 9 + k), 127 + 120 * np.cos(y / 7 - k), 255 * (x + y) / (w + h)], -1)
    return Image.fromarray(img.clip(0, 255).astype(np.uint8))


@pytest.mark.parametrize("size", [(64, 48), (50, 38), (512, 512), (16, 16)])
def test_mp4_round_trip(tmp_path, size):
    w, h = size
    frames = [_smooth(w, h, k) for k in range(3)] + [Image.new("RGB", size, (0, 0, 0)), Image.new("RGB", size, (255, 255, 255))]
    path = mp4.write_mp4(frames, str(tmp_path / "v.mp4"), fps=8)
    out, fps = mp4.read_mp4(path)
    assert fps == 8.0 and len(out) == len(frames) and all(o.size == size for o in out)
    for a, b in zip(frames, out):
        e = np.abs(np.asarray(a, dtype=np.int16) - np.asarray(b, dtype=np.int16))
        assert e.mean() < 5.0 and np.percentile(e, 99) <= 25  # 4:2:0 chroma + limited-range rounding on smooth content
    # lossless in YUV: decoding and re-encoding reproduces the file byte for byte apart from colour-conversion rounding
    y0 = mp4._rgb_to_yuv420(np.asarray(out[0]))[0]
    y1 = mp4._rgb_to_yuv420(np.asarray(mp4.read_mp4(mp4.write_mp4(out, str(tmp_path / "w.mp4"), fps=8))[0][0]))[0]
    assert np.abs(y0.astype(np.int16) - y1.astype(np.int16)).max() <= 1


def test_mp4_structure_and_headers(tmp_path):
    frames = [_smooth(50, 38, k) for k in range(4)]
    path = mp4.write_mp4(frames, str(tmp_path / "v.mp4"), fps=12)
    buf = open(path, "rb").read()
    top = [k for k, _, _ in mp4._boxes(buf, 0, len(buf))]
    assert top == [b"ftyp", b"moov", b"mdat"]          # moov before mdat: playable while downloading
    stbl = mp4._find(buf, 0, len(buf), b"moov", b"trak", b"mdia", b"minf", b"stbl")
    kinds = [k for k, _, _ in mp4._boxes(buf, *stbl)]
    assert kinds == [b"stsd", b"stts", b"stsc", b"stsz", b"stco"]
    sps = mp4._parse_sps(mp4._sps(50, 38))
    assert (sps["mbw"], sps["mbh"]) == (4, 3) and sps["crop"] == (0, 14, 0, 10)   # 64 x 48 coded, cropped to 50 x 38
    pps = mp4._parse_pps(mp4._pps())
    assert pps == dict(cabac=0, bottom=0, deblock=1, redundant=0)
    # every sample is one length-prefixed IDR NAL unit without start-code emulation
    co = mp4._find(buf, *stbl, b"stco")
    off = struct.unpack(">I", buf[co[0] + 8:co[0] + 12])[0]
    for _ in frames:
        ln = struct.unpack(">I", buf[off:off + 4])[0]
        nal = buf[off + 4:off + 4 + ln]
        assert nal[0] == 0x65 and b"\x00\x00\x00" not in nal and b"\x00\x00\x01" not in nal and b"\x00\x00\x02" not in nal
        off += 4 + ln
    assert off == len(buf)


def test_emulation_prevention_round_trip():
    rng = np.random.default_rng(0)
    raw = bytes(rng.integers(0, 4, 20000, dtype=np.uint8))  # dense in 00 00 0x patterns
    esc = mp4._escape(raw)
    assert mp4._unescape(esc) == raw
    assert all(esc[i:i + 2] != b"\x00\x00" or esc[i + 2] > 3 or esc[i + 2] == 3 for i in range(len(esc) - 2))
    assert b"\x00\x00\x00" not in esc and b"\x00\x00\x01" not in esc and b"\x00\x00\x02" not in esc


def test_foreign_streams_fail_with_the_reason(tmp_path):
    frames = [_smooth(32, 32, 0)]
    path = mp4.write_mp4(frames, str(tmp_path / "v.mp4"), fps=8)
    buf = bytearray(open(path, "rb").read())
    # flip macroblock 0's type from I_PCM to an intra-predicted one: what any real encoder's output looks like
    stbl = mp4._find(bytes(buf), 0, len(buf), b"moov", b"trak", b"mdia", b"minf", b"stbl")
    co = mp4._find(bytes(buf), *stbl, b"stco")
    off = struct.unpack(">I", buf[co[0] + 8:co[0] + 12])[0]
    nal0 = off + 4
    # slice header of write_mp4: ue(0) ue(7) ue(0) u4(0) ue(id) 00 se(0) ue(1) = 1 0001000 1 0000 1 00 1 010 | then ue(25)
    bits = np.unpackbits(np.frombuffer(bytes(buf[nal0 + 1:nal0 + 6]), dtype=np.uint8))
    pos = 1 + 7 + 1 + 4 + 1 + 2 + 1 + 3
    assert list(bits[pos:pos + 9]) == [0, 0, 0, 0, 1, 1, 0, 1, 0]  # ue(25)
    bits[pos:pos + 9] = [0, 0, 0, 0, 1, 1, 0, 0, 1]                # ue(24): I_16x16_3_2_1
    buf[nal0 + 1:nal0 + 6] = np.packbits(bits).tobytes()
    bad = tmp_path / "bad.mp4"
    bad.write_bytes(bytes(buf))
    with pytest.raises(mp4.Mp4Unsupported, match="I_PCM"):
        mp4.read_mp4(str(bad))
    with pytest.raises(RuntimeError, match="%05d.png"):
        utils.convert_video_to_frames(str(bad), (32, 32), save_frames=False)
    junk = tmp_path / "junk.mp4"
    junk.write_bytes(b"not an mp4 file at all")
    with pytest.raises(RuntimeError, match="cannot decode"):
        utils.convert_video_to_frames(str(junk), (32, 32), save_frames=False)


def test_convert_video_to_frames_follows_the_reference(tmp_path):
    """``i2vgen-xl/utils.py:43-67``: frames resized (LANCZOS) to img_size and saved as <dir>/<name>/%05d.png."""
    frames = [_smooth(96, 64, k) for k in range(3)]
    path = utils.export_to_video(frames, str(tmp_path / "clip.mp4"), fps=8)
    out = utils.convert_video_to_frames(path, img_size=(48, 32), save_frames=True)
    assert len(out) == 3 and all(o.size == (48, 32) for o in out)
    assert sorted(os.listdir(tmp_path / "clip")) == ["00000.png", "00001.png", "00002.png"]
    ref = frames[1].resize((48, 32), resample=Image.Resampling.LANCZOS)
    e = np.abs(np.asarray(ref, dtype=np.int16) - np.asarray(Image.open(tmp_path / "clip" / "00001.png"), dtype=np.int16))
    assert e.mean() < 5.0
