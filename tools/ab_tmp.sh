for seg in 0 64 256; do for v in abl119 base; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so python tools/bench_codecs.py --codec vorbis --steps 3 --vorbis-seg $seg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)['vorbis']; print('$v seg $seg', round(j['avg_kernel_ms'],3), round(j['frac'],4), j['bitwise_mismatches'])
"; done; done
