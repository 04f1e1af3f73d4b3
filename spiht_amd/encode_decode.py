"""Command-line round trip of one image file on the GPU: `python -m spiht_amd.encode_decode IMAGE [--bpp 0.1 ...]`.

The argument schema is the reference tool's (encode_decode.py:17-26: same names, defaults and meaning), so a command
line written for it runs here; `--save FILE` additionally writes the encoding (utils.save_encoding) and `--load FILE`
decodes such a file instead of encoding IMAGE.  The body is organised as three steps -- plan (settings, level and bit
budget from the arguments and the picture), encode or load, decode and report -- each of which is usable on its own.
"""
import math
import time
from argparse import ArgumentParser
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .spiht_wrapper import EncodingResult, SpihtSettings, decode_image, encode_image, get_slices_and_h_w
from .utils import imload, imsave, load_encoding, save_encoding

_ARGS = [  # (flag, type, default, help): the reference tool's options, then ours
    ("--bpp", float, 0.1, "bits per pixel"),
    ("--quantization_scale", float, 255.0, None),
    ("--level", int, None, "wavedec2 level. default is set so that the highest DWT level has a width and height of 4."),
    ("--wavelet", str, "bior2.2", "wavedec2 wavelet"),
    ("--mode", str, "reflect", "wavedec2 mode"),
    ("--color_model", str, "IPT", None),
    ("--per_channel_quant_scales", str, "1., 0.2, 0.2", None),
    ("--out", str, "reconstructed.png", "save reconstructed image to this file path"),
    ("--save", str, None, "also write the encoding to this file"),
    ("--load", str, None, "decode this encoding instead of encoding the image"),
]


def build_parser():
    parser = ArgumentParser(description=__doc__.splitlines()[0])
    parser.add_argument("image_filename")
    for flag, typ, default, text in _ARGS:
        parser.add_argument(flag, type=typ, default=default, help=text)
    return parser


def default_level(h, w):
    """the deepest level that leaves the coarsest band at least 8 samples on its short side (encode_decode.py:33-38)"""
    return math.floor(min(math.log2(h / 8), math.log2(w / 8)))


@dataclass
class Plan:
    settings: SpihtSettings
    level: int
    max_bits: int


def plan(args, c, h, w) -> Plan:
    """what to code the c x h x w picture with.  A grey picture takes neither a colour model nor three channel
    scales (the reference's defaults would raise on it)."""
    scales = [float(v) for v in args.per_channel_quant_scales.split(",")]
    colour = c == 3
    return Plan(
        settings=SpihtSettings(wavelet=args.wavelet, quantization_scale=args.quantization_scale, mode=args.mode,
                               color_model=args.color_model if colour else None,
                               per_channel_quant_scales=scales if len(scales) == c else None),
        level=default_level(h, w) if args.level is None else args.level,
        max_bits=round(args.bpp * h * w))  # encode_decode.py:43


def timed(fn, *a):
    t0 = time.perf_counter()
    out = fn(*a)
    return out, time.perf_counter() - t0


def report_encoding(enc: EncodingResult, settings: SpihtSettings, seconds: Optional[float]):
    if seconds is not None:
        print("encoded in %.3f s: %.2f KiB" % (seconds, len(enc.encoded_bytes) / 1024))
    slices, _, _ = get_slices_and_h_w(enc.h, enc.w, settings, enc.level)
    print("  level %s, start plane %d, coarsest band %d x %d" % (enc.level, enc.max_n, slices[0][1].stop, slices[0][2].stop))


def main(args):
    picture = imload(args.image_filename)
    c, h, w = picture.shape
    p = plan(args, c, h, w)
    if args.load:
        enc, secs = load_encoding(args.load), None
    else:
        print("encoding %d x %d x %d at %.3f bpp" % (c, h, w, args.bpp))
        enc, secs = timed(encode_image, picture, p.settings, p.level, p.max_bits)
    report_encoding(enc, p.settings, secs)
    if args.save:
        save_encoding(args.save, enc)
        print("  encoding written to", args.save)
    decoded, secs = timed(decode_image, enc, p.settings)
    decoded = np.asarray(decoded)[:, :h, :w]  # the inverse transform of an odd-sized picture is one sample longer
    print("decoded in %.3f s, mean squared error %.5f" % (secs, float(((picture - decoded) ** 2).mean())))
    imsave(args.out, decoded)
    print("  picture written to", args.out)
    return enc, decoded


if __name__ == "__main__":
    main(build_parser().parse_args())
