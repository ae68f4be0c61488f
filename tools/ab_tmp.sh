run() { env $2 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['ms_per_step'],2), round(d['ms_per_step_median'],2), round(d['config']['collection_s']*1e3,2), round(d['config']['learn_s']*1e3,2), round(d['value']/1e6,3))"; }
run mixed ""
run all128 "HX_BG_TILE=128"
run mixed ""
run all128 "HX_BG_TILE=128"
