import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'audio-formats_amd')
import numpy as np, afgpu, opus_bitstream as ob, oraclelib, torch
rng = np.random.default_rng(21)
gpu=torch.device("cuda:0")
for k in range(24):
    ch = 1 + k % 2
    data, _ = ob.random_celt_file(rng, ch, int(rng.integers(1, 70)), preskip=int(rng.integers(0, 121)), gain=(0, -1500, 700)[k % 3])
    if k not in (0, 8): continue
    rec = oraclelib.opus_decode_file(data)
    base, recs = oraclelib.opus_channel_records(rec)
    total = rec["pcm_frames"]*ch
    want = oraclelib.celt_transform(base, recs, rec["coeffs"], total)
    for path in ("split","stream"):
        import os; os.environ["AFG_CELT_PATH"]=path
        d_out = torch.full((total,), float("nan"), dtype=torch.float32, device=gpu)
        afgpu.celt_transform(len(base)-1, torch.from_numpy(base.view(np.int64)).to(gpu), torch.from_numpy(recs.view(np.uint8).copy()).to(gpu), torch.from_numpy(rec["coeffs"]).to(gpu), d_out, None)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        bad = np.nonzero(got.view(np.uint32)!=want.view(np.uint32))[0]
        print(k, path, "transform mismatches", len(bad), "of", total, "max abs want", np.abs(want).max())
        if len(bad):
            fr=rec["frames"]; starts=fr["out_off"]//ch
            for b in bad[:6]:
                fi=int(np.searchsorted(starts,b//ch,side="right")-1)
                print("   sample",b,"frame",fi,"fs",fr["frame_size"][fi],"blk",fr["blocks"][fi],"pf",fr["pf_period_new"][fi],fr["pf_gains_new"][fi],"got",got[b],"want",want[b], "in-frame", b//ch-starts[fi])
            fi=int(np.searchsorted(starts,bad[0]//ch,side="right")-1)
            print("   frames around:", [(int(fr["frame_size"][j]),int(fr["blocks"][j]),int(fr["pf_period_new"][j]),float(fr["pf_gains_new"][j][0])) for j in range(max(0,fi-3),fi+2)])
