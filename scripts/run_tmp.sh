for args in "--no-cpu-baseline" "--no-cpu-baseline --no-big-configs" "--no-cpu-baseline"; do
python bench.py $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', '|', d['ms_per_step'], [(o['config'][:12], o['ms_per_step']) for o in d['other_configs']])"
done
