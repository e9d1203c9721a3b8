import sys,os
ROOT=os.getcwd()
sys.path[:0]=[ROOT+'/audio-formats_amd',ROOT+'/tests',ROOT+'/tools']
import numpy as np, torch, afgpu
from afgpu import synthetic
import bench_codecs as B
dev=torch.device('cuda:0')
d_frames,d_sub,res,n_frames,total,frames,subframes=synthetic.flac_batch_device(0xF1AC,4096,323,dev)
out=torch.empty(total,dtype=torch.int32,device=dev)
ms=B.time_launches(lambda: afgpu.flac_transform(n_frames,d_frames,d_sub,res,out,None),3,1)
print('full', sum(ms)/len(ms))
sf=subframes.copy(); sf['order']=0; sf['coef']=0; sf['use64']=0
d_sub0=torch.from_numpy(sf.view(np.uint8).copy()).to(dev)
ms=B.time_launches(lambda: afgpu.flac_transform(n_frames,d_frames,d_sub0,res,out,None),3,1)
print('order0 (copy+decorrelate only)', sum(ms)/len(ms))
# plain copy ceiling
a=torch.empty(total,dtype=torch.int32,device=dev)
ms=B.time_launches(lambda: a.copy_(res),3,1)
print('torch copy same bytes', sum(ms)/len(ms))
