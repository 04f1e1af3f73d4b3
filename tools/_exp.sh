for v in 20000 100000 500000 20000 100000 500000; do
SPIHT_IDWT_PF_MIN=$v timeout -k 10 300 python bench.py --steps 24 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); s=r['stages_ms_per_step_summed_over_streams']; print('pfmin $v', r['ms_per_step'], s['encode_lists'], s['decode_lists'], s['dwt_level1'], s['idwt_rest'], s['idwt_level1'])"
done
