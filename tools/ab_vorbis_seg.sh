for seg in 0 32 64 128; do python tools/bench_codecs.py --codec vorbis --steps 3 --vorbis-seg $seg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)['vorbis']; print('seg $seg', round(j['avg_kernel_ms'],3), round(j['frac'],4), j['bitwise_mismatches'])
"; done
