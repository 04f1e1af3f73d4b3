"""Host helpers around the codec (reference: spiht/utils.py:6-20, encode_decode.py:17-90, wrapper:83-89)."""
import numpy as np
import pytest

from spiht_amd import utils
from spiht_amd.encode_decode import build_parser, default_level
from spiht_amd.spiht_wrapper import ENCODER_DECODER_VERSION, EncodingResult


def test_bytes_to_bits_is_lsb_first():
    bits = utils.bytes_to_bits(bytes([0x01, 0x80, 0xA5]))
    assert bits.tolist() == [1, 0, 0, 0, 0, 0, 0, 0,  0, 0, 0, 0, 0, 0, 0, 1,  1, 0, 1, 0, 0, 1, 0, 1]


def test_container_round_trip(tmp_path):
    enc = EncodingResult(bytes(range(256)) * 3 + b"\x00\xff", 1080, 1920, 3, 12, None)
    p = tmp_path / "a.spiht"
    utils.save_encoding(p, enc)
    back = utils.load_encoding(p)
    assert back == enc and back._encoding_version == ENCODER_DECODER_VERSION and isinstance(back.encoded_bytes, bytes)
    enc7 = EncodingResult(b"", 5, 7, 1, 0, 7)
    utils.save_encoding(p, enc7)
    assert utils.load_encoding(p) == enc7


def test_container_rejects_garbage(tmp_path):
    p = tmp_path / "bad.spiht"
    p.write_bytes(b"PNG\x00 not a container")
    with pytest.raises(ValueError):
        utils.load_encoding(p)
    p.write_bytes(b"SPHT" + (1000).to_bytes(4, "little") + b"{}")
    with pytest.raises(ValueError):
        utils.load_encoding(p)


def test_imload_imsave(tmp_path):
    pytest.importorskip("PIL")
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (3, 9, 14)).astype(np.uint8)
    utils.imsave(tmp_path / "c.png", rgb / 255)
    got = utils.imload(tmp_path / "c.png")
    assert got.shape == (3, 9, 14) and got.dtype == np.float64 and np.array_equal(np.round(got * 255).astype(np.uint8), rgb)
    grey = rng.integers(0, 256, (1, 6, 5)).astype(np.uint8)
    utils.imsave(tmp_path / "g.png", grey / 255)
    got = utils.imload(tmp_path / "g.png")
    assert got.shape == (1, 6, 5) and np.array_equal(np.round(got * 255).astype(np.uint8), grey)


def test_cli_defaults_match_reference():
    a = build_parser().parse_args(["x.png"])
    assert (a.bpp, a.quantization_scale, a.level, a.wavelet, a.mode, a.color_model, a.per_channel_quant_scales, a.out) == \
        (0.1, 255.0, None, "bior2.2", "reflect", "IPT", "1., 0.2, 0.2", "reconstructed.png")
    assert default_level(1080, 1920) == 7 and default_level(512, 512) == 6 and default_level(64, 4096) == 3


def test_reference_import_name():
    """An unchanged caller of the reference says `import spiht` (reference encode_decode.py:10-14, make_gif.py:9-12,
    demonstrate.py:10-13): the alias package hands it this repository's modules, the same objects."""
    import spiht
    import spiht_amd
    from spiht import encode_image, decode_image, SpihtSettings, EncodingResult, ENCODER_DECODER_VERSION  # noqa: F401
    from spiht.spiht_wrapper import get_slices_and_h_w  # noqa: F401
    from spiht.utils import imload  # noqa: F401
    from spiht.spiht import decode, encode
    import spiht.spiht as ext
    assert ext is spiht_amd.spiht and spiht.spiht is ext
    assert encode_image is spiht_amd.encode_image and decode is spiht_amd.decode and encode is spiht_amd.encode
    assert spiht.spiht_wrapper is spiht_amd.spiht_wrapper and SpihtSettings is spiht_amd.SpihtSettings
