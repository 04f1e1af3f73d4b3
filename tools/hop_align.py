#!/usr/bin/env python3
"""Build helper (spiht_amd/csrc/Makefile): how many s_nop to put in front of the decoder's window loop so that the
hand-written hop loop of k_decode<false> starts 16 bytes into a 32-byte block -- measured best on MI355X with the
11-instruction hop body, four hops per loop trip (decode of 256 images by offset: 16: 6.98 ms, 8: 7.07, 20: 7.10,
0: 7.13, 28: 7.17, 24: 7.18, 4: 7.24, 12: 7.25; with the earlier 12-instruction body the best offset was 20).
Input: decode.hip compiled for the device only with -DDEC_PAD=0.  Prints the pad count (0 if anything goes wrong: the
build then just keeps whatever alignment it has)."""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/llvm/bin/"
try:
    obj = sys.argv[1]
    elf = tempfile.mktemp(suffix=".elf", dir="/tmp")
    subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dis = subprocess.check_output([LLVM + "llvm-objdump", "-d", elf], text=True)
    # the hop loop is the only user of s_ff1_i32_b64 followed by s_add_i32 / two s_lshr_b64; k_decode<false> comes last
    addrs = [int(m.group(1), 16) for m in re.finditer(r"s_ff1_i32_b64 .*// ([0-9A-Fa-f]+):", dis)]
    BODIES = 4         # hop bodies per loop trip (decode.hip)
    label = addrs[-BODIES]  # the first one of the last group is the loop label
    print(((16 - label) % 32) // 4)
except Exception:
    print(0)
