for cfg in "20 3" "100 3" "500 3" "20 100" "20 3" "1000 10"; do set -- $cfg
  python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps',d['steps'],'warmup',d['warmup'],'value',round(d['value']),'ms',round(d['ms_per_step'],4),'median',round(d['ms_per_step_event_median'],4),{k:round(v,3) for k,v in d['roofline']['avg_kernel_ms'].items()})"
done
