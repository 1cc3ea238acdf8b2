"""Checkpoint files for the engines: a dependency-free reader / writer of the safetensors container (8-byte little-endian header
length, JSON header {name: {dtype, shape, data_offsets}}, raw little-endian tensor bytes) — the format of the files the reference
pulls by model name (`unet/diffusion_pytorch_model.safetensors`, `vae/...`; src/stable_diffusion_depth.py:58-88 `from_pretrained`).
Offline there are no such files, so the reader is exercised on files the tests write themselves (and cross-checked against the
`safetensors` package where that is importable).  Nothing in a file is executed: the header is JSON, the payload is raw numbers.

    sd = load_file("unet/diffusion_pytorch_model.safetensors")     # {name: torch.Tensor (CPU, memory-mapped)}
    UNet2DConditionModel.from_file(path, device=...)               # engine with its weights repacked from the file
"""
import json
import os
import struct
import warnings
import numpy as np
import torch

_DTYPES = {"F64": (np.float64, torch.float64), "F32": (np.float32, torch.float32), "F16": (np.float16, torch.float16),
           "BF16": (np.uint16, torch.bfloat16), "I64": (np.int64, torch.int64), "I32": (np.int32, torch.int32),
           "I16": (np.int16, torch.int16), "I8": (np.int8, torch.int8), "U8": (np.uint8, torch.uint8), "BOOL": (np.uint8, torch.bool)}
_NAMES = {v[1]: k for k, v in _DTYPES.items()}
MAX_HEADER = 100 * 1024 * 1024


class SafetensorsError(ValueError):
    pass


def read_header(path):
    """-> (header dict without __metadata__, metadata dict, offset of the data section)."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        raw = f.read(8)
        if len(raw) != 8:
            raise SafetensorsError(f"{path}: shorter than a safetensors header")
        (n,) = struct.unpack("<Q", raw)
        if n > MAX_HEADER or 8 + n > size:
            raise SafetensorsError(f"{path}: header length {n} does not fit the file ({size} bytes)")
        try:
            hdr = json.loads(f.read(n).decode("utf-8"))
        except (UnicodeDecodeError, json.JSONDecodeError) as e:
            raise SafetensorsError(f"{path}: header is not JSON: {e}")
    if not isinstance(hdr, dict):
        raise SafetensorsError(f"{path}: header is not an object")
    meta = hdr.pop("__metadata__", {}) or {}
    data0, avail = 8 + n, size - 8 - n
    for name, e in hdr.items():
        if not isinstance(e, dict) or e.get("dtype") not in _DTYPES or "shape" not in e or "data_offsets" not in e:
            raise SafetensorsError(f"{path}: bad entry for {name!r}")
        b, en = e["data_offsets"]
        want = int(np.prod(e["shape"], dtype=np.int64)) * np.dtype(_DTYPES[e["dtype"]][0]).itemsize
        if not (0 <= b <= en <= avail) or en - b != want:
            raise SafetensorsError(f"{path}: {name!r} claims bytes [{b}, {en}) for shape {e['shape']} {e['dtype']} ({want} bytes)")
    return hdr, meta, data0


def load_file(path, names=None):
    """-> {name: CPU tensor}.  Tensors are views of ONE read-only memory map of the file (nothing is copied until the engine's
    repack kernels read them), so a multi-GB checkpoint costs no host memory beyond the page cache."""
    hdr, _, data0 = read_header(path)
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    out = {}
    for name, e in hdr.items():
        if names is not None and name not in names:
            continue
        npdt, tdt = _DTYPES[e["dtype"]]
        b, en = e["data_offsets"]
        arr = np.frombuffer(mm, dtype=npdt, count=(en - b) // np.dtype(npdt).itemsize, offset=data0 + b).reshape(e["shape"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", UserWarning)         # torch warns that a read-only map is not writable: it is never written
            t = torch.from_numpy(arr)
        if tdt == torch.bfloat16:
            t = t.view(torch.bfloat16)
        elif tdt == torch.bool:
            t = t.view(torch.bool)
        out[name] = t
    return out


def save_file(tensors, path, metadata=None):
    """Writer (tests, `state_dict` export): entries sorted by name, data 8-byte aligned as the format's reference writer does."""
    hdr, off, blobs = {}, 0, []
    for name in sorted(tensors):
        t = tensors[name].detach().cpu().contiguous()
        if t.dtype not in _NAMES:
            raise SafetensorsError(f"{name}: dtype {t.dtype} has no safetensors name")
        raw = (t.view(torch.uint16) if t.dtype == torch.bfloat16 else t.view(torch.uint8) if t.dtype == torch.bool else t).numpy().tobytes()
        hdr[name] = {"dtype": _NAMES[t.dtype], "shape": list(t.shape), "data_offsets": [off, off + len(raw)]}
        off += len(raw)
        blobs.append(raw)
    if metadata:
        hdr["__metadata__"] = {str(k): str(v) for k, v in metadata.items()}
    js = json.dumps(hdr, separators=(",", ":")).encode("utf-8")
    js += b" " * ((8 - len(js) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(js)))
        f.write(js)
        for raw in blobs:
            f.write(raw)
