import sys; sys.path[:0]=['tests','audio-formats_amd']
import numpy as np, torch, oraclelib, afgpu
from afgpu import VorbisPlan, synthetic, VORBIS_LONG, VORBIS_PREV, VORBIS_NEXT
dev=torch.device('cuda:0')
L=VORBIS_LONG; P=VORBIS_PREV; N=VORBIS_NEXT
pflags=np.array([L|P|N, L|P, 0, 0, L|N, L|P|N], np.uint8)
npk=len(pflags)
for C in (1,2):
    rng=np.random.default_rng(0)
    plan=VorbisPlan([npk],[C],[256],[2048],pflags,1000)
    spec=rng.standard_normal(plan.spec_floats).astype(np.float32)
    so,oo=plan.offsets()
    want=oraclelib.vorbis_transform([npk],[C],[256],[2048],pflags,so,oo,spec,plan.out_floats)
    d_out=torch.full((plan.out_floats,),float('nan'),dtype=torch.float32,device=dev)
    plan.transform(torch.from_numpy(spec).to(dev), d_out); torch.cuda.synchronize()
    got=d_out.cpu().numpy()
    for p in range(1,npk):
        seg=slice(int(oo[p]), int(oo[p+1]) if p+1<npk else got.size)
        b=(got[seg].view(np.uint32)!=want[seg].view(np.uint32))
        print('C',C,'pkt',p,'flags',pflags[p],'bad',b.sum(),'of',b.size,'first bad',np.flatnonzero(b)[:4],'nan',np.isnan(got[seg]).sum())
