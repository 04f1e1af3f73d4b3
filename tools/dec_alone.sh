#!/bin/bash
# the list decoder alone (HIP events around the kernel) for each library given: 4 copies of the bench's first image, the
# 256-image batch, and one 4096 x 4096 picture at 1 bpp
for lib in "$@"; do
  for n in 4 256; do
    echo -n "$lib  B=$n  "
    SPIHT_HIP_LIB=$lib timeout -k 10 300 python tools/prof_decode.py $n 2>&1 | grep "decoder kernel"
  done
  echo -n "$lib  4096^2 1 bpp  "
  PROF_H=4096 PROF_W=4096 PROF_LEVEL=9 PROF_WAVELET=bior6.8 PROF_BPP=1.0 SPIHT_HIP_LIB=$lib timeout -k 10 300 python tools/prof_decode.py 1 2>&1 | grep "decoder kernel"
done
