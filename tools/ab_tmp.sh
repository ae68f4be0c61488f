run() { env $2 python bench.py --no-cpu-baseline $3 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', round(d['ms_per_step'],2), round(d['ms_per_step_median'],2), round(d['config']['collection_s']*1e3,2), round(d['config']['learn_s']*1e3,2), round(d['value']/1e6,3), d['config']['observation_storage'])"; }
run rows "" "--storage rows"
run frames "" "--storage frames"
run rows "" "--storage rows"
run frames "" "--storage frames"
run frames_gather "HX_FRAMES_GATHER=1" "--storage frames"
run frames_chunk3 "HX_CRITIC_CHUNK=3" "--storage frames"
run frames_tile64 "HX_BG_TILE=64" "--storage frames"
