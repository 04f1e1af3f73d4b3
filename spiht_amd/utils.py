"""Host helpers of the reference's `spiht/utils.py` that its callers use around the codec (bit view of a stream,
image loading), plus a small container for an `EncodingResult` so that the command-line tool can write what it
encoded.  Plotting helpers (matplotlib) of the reference are not part of this package.
"""
import json
import struct

import numpy as np

from .spiht_wrapper import EncodingResult


def bytes_to_bits(spiht_bytes: bytes):
    """utils.py:6-9: the stream as a 0/1 array, least significant bit of every byte first (bitvec Lsb0 order)."""
    np_bytes = np.frombuffer(spiht_bytes, np.uint8)
    return np.unpackbits(np_bytes, bitorder='little')


def imload(path) -> np.ndarray:
    """utils.py:12-20: image file -> float64 (C,H,W) in [0,1]; a grey image gets a leading axis of 1."""
    from PIL import Image  # optional dependency: only the file helpers need it
    im = np.asarray(Image.open(path))
    if im.ndim > 2:
        im = np.moveaxis(im, -1, 0)
    else:
        im = im[None, :, :]
    return im / 255


def imsave(path, im) -> None:
    """(C,H,W) float image in [0,1] -> 8-bit image file, as encode_decode.py:76-84 does."""
    from PIL import Image
    im = np.asarray(im)
    im = im[0] if im.shape[0] == 1 else np.moveaxis(im, 0, -1)
    Image.fromarray((im.clip(0.0, 1.0) * 255).astype(np.uint8)).save(path)


_MAGIC = b"SPHT"


def save_encoding(path, enc: EncodingResult) -> None:
    """Container: magic, u32 length of a JSON header (EncodingResult.to_dict() without the stream, wrapper:83-84),
    the header, then the stream bytes."""
    d = enc.to_dict()
    stream = d.pop("encoding_result_encoded_bytes")
    head = json.dumps(d).encode()
    with open(path, "wb") as f:
        f.write(_MAGIC + struct.pack("<I", len(head)) + head + bytes(stream))


def load_encoding(path) -> EncodingResult:
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:4] != _MAGIC or len(blob) < 8:
        raise ValueError("not a SPIHT container: %r" % (path,))
    (n,) = struct.unpack("<I", blob[4:8])
    if 8 + n > len(blob):
        raise ValueError("truncated SPIHT container: %r" % (path,))
    d = json.loads(blob[8:8 + n].decode())
    d["encoding_result_encoded_bytes"] = blob[8 + n:]
    return EncodingResult.from_dict(d)  # decode_image checks the version field (wrapper:226-227)
