python bench.py --no-big-configs --no-encoder-probe --steps 6 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('default lib', '|', d['ms_per_step'], [(o['config'][:14], o['ms_per_step']) for o in d['other_configs']])"
