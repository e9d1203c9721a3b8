#!/bin/bash
# tools/gpu_c5_order.sh: BASELINE configs[4] on one GPU with the Opus members overlapped / serial and different launch orders
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
cd "$R"; mkdir -p gpurun_out
run() { python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'order':'$1','ms_per_step':d['ms_per_step'],'value':d['value'],'kernels':{k['codec']:round(k['avg_kernel_ms'],2) for k in d['roofline']['kernels']},'parity_ok':all(p['mismatches']==0 for p in d['parity'].values())}))
"; }
( run default
  AFG_C5_ORDER=serial run serial
  AFG_C5_ORDER=celt,vorbis,mp3,flac run celt,vorbis,mp3,flac
  AFG_C5_ORDER=celt,flac,mp3,vorbis run celt,flac,mp3,vorbis
  AFG_C5_ORDER=mp3,vorbis,flac,celt run mp3,vorbis,flac,celt
  AFG_C5_ORDER=celt,flac,vorbis,mp3 run celt,flac,vorbis,mp3 ) | tee gpurun_out/r04_c5_order.jsonl
