cd $GRAFT_REPO_ROOT
for t in 0 32 48 64; do python tools/bench_codecs.py --codec opus_e2e --e2e-threads $t --e2e-distinct 64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        v=list(json.loads(l).values())[0]; print('opus threads=$t', round(v['samples_per_s_end_to_end']/1e9,2), round(v['seconds']*1e3,1), v['parity']['mismatches'])
"; done
for c in mp3_e2e vorbis_e2e flac_e2e; do python tools/bench_codecs.py --codec $c --e2e-distinct 64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        v=list(json.loads(l).values())[0]; print('$c default', round(v['samples_per_s_end_to_end']/1e9,2), round(v['seconds']*1e3,1), v['parity']['mismatches'])
"; done
