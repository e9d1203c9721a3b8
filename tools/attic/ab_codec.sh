#!/bin/bash
# tools/ab_codec.sh <codec> <variant>... : time one codec leg of tools/bench_codecs.py per library variant
codec=$1; shift
for v in "$@"; do AFG_LIB_PATH=$PWD/audio-formats_amd/lib/libafg_$v.so python tools/bench_codecs.py --codec $codec --steps 3 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l)['$codec']; print('$v', round(j['avg_kernel_ms'],3), round(j['frac'],4), {k:v for k,v in j.items() if 'mismatch' in k})
"; done
