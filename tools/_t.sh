python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for w in sign65 sign87 sign44; do
python bench.py --workload $w --no-extras --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', '%.4g'%d['value'], '%.4f'%d['ms_per_step'], {k:round(v,3) for k,v in d.get('stage_ms_per_step',{}).items()})"
done
