"""Encode / decode a single image file on the GPU: the counterpart of the reference's `encode_decode.py` (:17-90,
same arguments and defaults).  `python -m spiht_amd.encode_decode IMAGE [--bpp 0.1 ...]`

Two additions: `--save FILE` writes the encoding (utils.save_encoding), `--load FILE` decodes such a file instead
of encoding IMAGE.
"""
import math
import time
from argparse import ArgumentParser

import numpy as np

from .spiht_wrapper import SpihtSettings, decode_image, encode_image, get_slices_and_h_w
from .utils import imload, imsave, load_encoding, save_encoding


def build_parser():
    parser = ArgumentParser()
    parser.add_argument('image_filename')
    parser.add_argument('--bpp', help='bits per pixel', type=float, default=0.1)
    parser.add_argument('--quantization_scale', default=255.0, type=float)
    parser.add_argument('--level', help='wavedec2 level. default is set so that the highest DWT level has a width '
                                        'and height of 4.', default=None, type=int)
    parser.add_argument('--wavelet', help='wavedec2 wavelet', default='bior2.2', type=str)
    parser.add_argument('--mode', help='wavedec2 mode', default='reflect', type=str)
    parser.add_argument('--color_model', default="IPT", type=str)
    parser.add_argument('--per_channel_quant_scales', default="1., 0.2, 0.2", type=str)
    parser.add_argument('--out', help='save reconstructed image to this file path', type=str, default='reconstructed.png')
    parser.add_argument('--save', help='also write the encoding to this file', type=str, default=None)
    parser.add_argument('--load', help='decode this encoding instead of encoding the image', type=str, default=None)
    return parser


def default_level(h, w):
    """encode_decode.py:33-38"""
    return math.floor(min(math.log2(h / 8), math.log2(w / 8)))


def main(args):
    im = imload(args.image_filename)
    c, h, w = im.shape
    level = default_level(h, w) if args.level is None else args.level
    max_bits = round(args.bpp * h * w)
    per_channel_quant_scales = list(float(x) for x in args.per_channel_quant_scales.split(","))
    if c != len(per_channel_quant_scales):  # a grey image cannot take three channel scales, nor a colour model
        per_channel_quant_scales = None
    spiht_settings = SpihtSettings(
        quantization_scale=args.quantization_scale,
        mode=args.mode,
        wavelet=args.wavelet,
        color_model=args.color_model if c == 3 else None,
        per_channel_quant_scales=per_channel_quant_scales,
    )
    if args.load:
        encoded = load_encoding(args.load)
    else:
        print(f"Starting encoding of image {c} {h} {w}")
        st = time.time()
        encoded = encode_image(im, spiht_settings, level, max_bits)
        et = time.time()
        print(f"Encoding done in {et-st:.3f}s. Image encoded to {len(encoded.encoded_bytes) / 1024:.2f}kb")
    print(f"   levels: {encoded.level}")
    print(f"    max n: {encoded.max_n}")
    slices, enc_h, enc_w = get_slices_and_h_w(encoded.h, encoded.w, spiht_settings, encoded.level)
    ll_h, ll_w = slices[0][1].stop, slices[0][2].stop
    print(f"ll_h ll_w: {ll_h, ll_w}")
    if args.save:
        save_encoding(args.save, encoded)
        print("Encoding saved to ", args.save)
    st = time.time()
    dec_im = decode_image(encoded, spiht_settings)
    et = time.time()
    dec_im = np.asarray(dec_im)[:, :h, :w]  # the inverse transform of an odd-sized image is one sample longer
    print(f"Decoding done in {et-st:.3f}s. L2 distance: {((im-dec_im)**2).mean():.5f}")
    imsave(args.out, dec_im)
    print("Saved to ", args.out)
    return encoded, dec_im


if __name__ == "__main__":
    main(build_parser().parse_args())
