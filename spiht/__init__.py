"""`import spiht` for callers written against the reference (its own scripts say `from spiht import encode_image`,
`from spiht.spiht_wrapper import SpihtSettings`, `from spiht.spiht import decode`, `from spiht.utils import imload`:
/root/reference/encode_decode.py:10-14, make_gif.py:9-12, demonstrate.py:10-13).

The package of this repository is `spiht_amd`; this one only re-exports it under the reference's name, module by
module, so an unchanged caller picks up the MI355X path.  Nothing is implemented here.
"""
import importlib
import sys

import spiht_amd as _impl

_SUBMODULES = ("spiht_wrapper", "spiht", "utils", "color_models", "spiht_py", "encode_decode")
for _name in _SUBMODULES:
    # `spiht.spiht` is the reference's compiled extension (src/lib.rs:58-65); here spiht_amd/spiht.py over the C ABI
    _mod = importlib.import_module("spiht_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
del _name, _mod

globals().update({_k: getattr(_impl, _k) for _k in ("encode_image", "decode_image", "EncodingResult", "SpihtSettings",
                                                    "ENCODER_DECODER_VERSION", "encode", "decode")})
__all__ = ["encode_image", "decode_image", "EncodingResult", "SpihtSettings", "ENCODER_DECODER_VERSION", "encode", "decode"]
