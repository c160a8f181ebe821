"""Reader for ``.safetensors`` files that needs nothing but torch (role of ``lora_diffusion/safe_open.py:1-68``:
the import fallback of ``lora.py`` / ``lora_manager.py`` when the ``safetensors`` wheel is absent).

File layout (safetensors spec): ``u64 little-endian header length | JSON header | raw tensor bytes``; every header entry
is ``name -> {"dtype", "shape", "data_offsets": [begin, end)}`` relative to the end of the header, plus the optional
``"__metadata__"`` string map.  The handle mimics the three calls ``parse_safeloras*`` makes: ``metadata()``,
``keys()``, ``get_tensor(name)``; it also works as a context manager like the real one.
"""
from __future__ import annotations

import json
import struct
from typing import Dict

import torch

_DTYPE = {"F64": torch.float64, "F32": torch.float32, "F16": torch.float16, "BF16": torch.bfloat16,
          "I64": torch.int64, "I32": torch.int32, "I16": torch.int16, "I8": torch.int8, "U8": torch.uint8,
          "BOOL": torch.bool}


class SafeOpenHandle:
    """Lazy view of one file: the payload is read once, tensors are cut out of it on demand."""

    def __init__(self, filename: str, device="cpu"):
        with open(filename, "rb") as f:
            head = f.read(8)
            if len(head) != 8:
                raise ValueError(f"{filename}: not a safetensors file (shorter than its length prefix)")
            (n,) = struct.unpack("<Q", head)
            header = json.loads(f.read(n).decode("utf-8"))
            self._payload = bytearray(f.read())
        self._meta: Dict[str, str] = header.pop("__metadata__", None) or {}
        self._index = header
        self._device = device
        for name, info in header.items():
            b, e = info["data_offsets"]
            if not (0 <= b <= e <= len(self._payload)):
                raise ValueError(f"{filename}: tensor {name!r} points outside the file")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def metadata(self) -> Dict[str, str]:
        return self._meta

    def keys(self):
        return self._index.keys()

    def get_tensor(self, name: str) -> torch.Tensor:
        info = self._index[name]
        try:
            dt = _DTYPE[info["dtype"]]
        except KeyError:
            raise TypeError(f"safe_open: dtype {info['dtype']} of {name!r} is not supported") from None
        b, e = info["data_offsets"]
        shape = tuple(info["shape"])
        if e == b:
            return torch.empty(shape, dtype=dt, device=self._device)
        flat = torch.frombuffer(self._payload, dtype=torch.uint8, count=e - b, offset=b)
        return flat.clone().view(dt).reshape(shape).to(self._device)


def safe_open(filename, framework="pt", device="cpu") -> SafeOpenHandle:
    if framework != "pt":
        raise ValueError("`framework` must be 'pt'")
    return SafeOpenHandle(filename, device)
