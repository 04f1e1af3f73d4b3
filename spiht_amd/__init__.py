"""spiht_amd: MI355X-native drop-in for the hot path of theAdamColton/spiht.

Mirrors /root/reference/spiht/__init__.py:1-2: the wrapper API plus the two functions of the compiled
extension.  Importing the package loads libspiht_hip.so and fails loudly if it is absent.
"""
from . import _lib as _lib_mod

_lib_mod.lib()  # fail at import time, not at first use, when the HIP library has not been built

from .spiht_wrapper import (encode_image, decode_image, EncodingResult, SpihtSettings,  # noqa: E402,F401
                            ENCODER_DECODER_VERSION)
from .spiht import encode, decode  # noqa: E402,F401
