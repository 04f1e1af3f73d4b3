#!/bin/bash
# Where the list decoder's kernel reloads spilled registers (it is capped at 96 VGPRs, decode.hip: DEC_WAVES_PER_EU): compiles
# decode.hip with line tables and lists every scratch_load of k_decode<false> with the source line it belongs to.  A reload inside
# the workers' per-window loop (worker_phase / work_lis) or the sequencer's walk costs a trip to memory per window -- beside an
# HBM-bound kernel that is microseconds (round 4: +25 % on the whole kernel) -- so check this list after touching the kernel.
#   tools/spill_sites.sh [8|12]
set -e
W=${1:-8}
root=$(cd "$(dirname "$0")/.." && pwd)
F=""; K=_ZN8dec_main8k_decodeILb0E
if [ "$W" = 8 ]; then F="-DDEC_NW=8 -DDEC_VARIANT=w8"; K=_ZN6dec_w88k_decodeILb0E; fi
T=$(mktemp -d)
cp "$root"/spiht_amd/csrc/*.h "$T"/ && cp "$root/spiht_amd/csrc/decode.hip" "$T/decode.hip"
(cd "$T" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $F -DDEC_PAD=0 -gline-tables-only --cuda-device-only -S decode.hip -o d.s 2>/dev/null)
L0=$(grep -n "^$K" "$T/d.s" | cut -d: -f1)
awk -v a="$L0" 'NR>=a' "$T/d.s" | awk '/^\.Lfunc_end/{exit} {print}' > "$T/k.s"
grep -A8 "$K" "$T/d.s" | grep -m2 "vgpr_spill_count\|group_segment" || true
awk '/\.amdhsa_kernel '"$K"'/,/\.end_amdhsa_kernel/' "$T/d.s" | grep "group_segment_fixed_size\|next_free_vgpr" || true
awk '/\.loc/{loc=$0} /scratch_load/{print $1" "$2" "$3" "$4" "$5"  @ "loc}' "$T/k.s" | sed 's/\t/ /g;s/\.loc 0 //;s/ *; /  ; /' | cut -c1-150
rm -rf "$T"
