for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', round(j['ms_per_step'],3), round(j['roofline']['frac'],4), j['parity']['bitwise_mismatches'])
"; done
