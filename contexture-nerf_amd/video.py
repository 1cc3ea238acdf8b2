"""The orbit video of `evaluate(save_as_video=True)` (src/training/trainer.py:943-950: `imageio.mimsave(... '.mp4', fps=25, quality=8)`).
imageio / ffmpeg are not importable offline, so there is no H.264 encoder; the frames are muxed as Motion-JPEG in an AVI (RIFF) container
instead — every frame an independent JPEG (PIL), index chunk at the end — which the usual players and `ffmpeg -i x.avi x.mp4` read.
Host-side output code, nothing of it is on the painting path."""
import io
import struct


def _chunk(tag, payload):
    pad = b"\x00" if len(payload) & 1 else b""
    return tag + struct.pack("<I", len(payload)) + payload + pad


def _list(kind, payload):
    return b"LIST" + struct.pack("<I", len(payload) + 4) + kind + payload


def write_mjpeg_avi(path, frames, fps=25, quality=90):
    """frames: sequence of uint8 arrays [H, W, 3] (all the same size).  Returns the number of frames written."""
    import numpy as np
    from PIL import Image
    jpegs = []
    h = w = None
    for f in frames:
        f = np.ascontiguousarray(f)
        if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
            raise ValueError(f"write_mjpeg_avi: frames must be uint8 [H, W, 3], got {f.dtype} {f.shape}")
        if h is None:
            h, w = f.shape[:2]
        elif (h, w) != f.shape[:2]:
            raise ValueError("write_mjpeg_avi: frames differ in size")
        b = io.BytesIO()
        Image.fromarray(f).save(b, format="JPEG", quality=quality)
        jpegs.append(b.getvalue())
    if not jpegs:
        raise ValueError("write_mjpeg_avi: no frames")
    n, biggest = len(jpegs), max(len(j) for j in jpegs)
    usec = int(round(1e6 / fps))
    avih = struct.pack("<14I", usec, biggest * fps, 0, 0x10, n, 0, 1, biggest, w, h, 0, 0, 0, 0)       # 0x10: AVIF_HASINDEX
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, fps, 0, n, biggest, 0xFFFFFFFF, 0) + struct.pack("<4h", 0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)                # BITMAPINFOHEADER
    hdrl = _list(b"hdrl", _chunk(b"avih", avih) + _list(b"strl", _chunk(b"strh", strh) + _chunk(b"strf", strf)))
    movi_payload, index, off = b"", b"", 4                                                               # offsets are relative to 'movi'
    for j in jpegs:
        c = _chunk(b"00dc", j)
        index += b"00dc" + struct.pack("<III", 0x10, off, len(j))                                        # AVIIF_KEYFRAME
        movi_payload += c
        off += len(c)
    body = b"AVI " + hdrl + _list(b"movi", movi_payload) + _chunk(b"idx1", index)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return n


def read_mjpeg_avi(path):
    """The inverse, for tests and for tools that want the frames back: (fps, [uint8 H x W x 3 arrays])."""
    import numpy as np
    from PIL import Image
    d = open(path, "rb").read()
    if d[:4] != b"RIFF" or d[8:12] != b"AVI ":
        raise ValueError("not an AVI file")
    usec = struct.unpack_from("<I", d, d.index(b"avih") + 8)[0]
    movi = d.index(b"movi")
    idx = d.rindex(b"idx1")
    n = struct.unpack_from("<I", d, idx + 4)[0] // 16
    frames = []
    for i in range(n):
        tag, flags, off, size = struct.unpack_from("<4sIII", d, idx + 8 + 16 * i)
        start = movi + off + 8
        frames.append(np.asarray(Image.open(io.BytesIO(d[start:start + size])).convert("RGB")))
    return 1e6 / usec, frames
