"""The container the third-party pin harness exchanges arrays in (inputs.pin / pins.pin): a flat list of named arrays.

    file   := "RGBDPIN1" record*
    record := u32 name_len, name bytes (utf-8), u32 dtype_code, u32 ndim, u64 dims[ndim], raw little-endian data (C order)

dtype codes: 0 u8, 1 i32, 2 f32, 3 f64, 4 u16, 5 i64.  pinfile.hpp is the C++ twin."""
import struct

import numpy as np

MAGIC = b"RGBDPIN1"
CODES = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.float64, 4: np.uint16, 5: np.int64}
BY_DTYPE = {np.dtype(v): k for k, v in CODES.items()}


def write(path, arrays):
    """arrays: dict name -> array (written in insertion order)."""
    with open(path, "wb") as f:
        f.write(MAGIC)
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype not in BY_DTYPE:
                raise TypeError("%s: dtype %s has no pin code" % (name, a.dtype))
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<II", BY_DTYPE[a.dtype], a.ndim))
            f.write(struct.pack("<%dQ" % a.ndim, *a.shape))
            f.write(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())


def read(path):
    out = {}
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("%s is not a pin file" % path)
        while True:
            h = f.read(4)
            if not h:
                return out
            name = f.read(struct.unpack("<I", h)[0]).decode()
            code, ndim = struct.unpack("<II", f.read(8))
            dims = struct.unpack("<%dQ" % ndim, f.read(8 * ndim)) if ndim else ()
            dt = np.dtype(CODES[code]).newbyteorder("<")
            n = int(np.prod(dims, dtype=np.int64)) if ndim else 1
            out[name] = np.frombuffer(f.read(n * dt.itemsize), dtype=dt).reshape(dims).copy()
