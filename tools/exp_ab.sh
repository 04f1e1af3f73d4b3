#!/bin/bash
# A/B of environment-switched experiments on one box: tools/exp_ab.sh <log> "<env assignments>" ...   ("" = baseline)
out=$1; shift
: > $out
for e in "$@"; do
  echo "=== [$e]" >> $out
  env $e timeout -k 10 200 python bench.py --cpu-sample 0 --steps 10 --warmup 3 2>>$out | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        print('ms_per_step', d['ms_per_step'], 'bit-exact', d['check'].get('stream_bit_exact_vs_oracle'), d['check'].get('decoded_image_bit_exact_vs_oracle'), d['check'].get('nbits_all_equal_budget'))
        print(json.dumps(d.get('stages_ms_per_step_summed_over_streams')))
        print(json.dumps(d.get('single_image_latency')))
" >> $out
done
grep -v "synthesised" $out
