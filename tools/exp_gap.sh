#!/bin/bash
# round 4: the timer between the inverse level 1 and the next decoder (SPIHT_EXP_GAP_US) against the LDS pad alone
out=$1; shift
: > $out
python -c "
from spiht_amd import _lib
c=_lib.default_context(); print('num_cu', c.get_option('num_cu'), 'lds_per_cu', c.get_option('lds_per_cu'))" >> $out 2>&1
for e in "$@"; do
  echo "=== [$e]" >> $out
  env $e timeout -k 10 200 python bench.py --cpu-sample 0 --steps 10 --warmup 3 2>>$out | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        print('ms_per_step', d['ms_per_step'], 'single', d['single_image_latency'])
        print(json.dumps(d.get('stages_ms_per_step_summed_over_streams')))
" >> $out
done
cat $out
