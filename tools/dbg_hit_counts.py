import sys, numpy as np, torch
sys.path.insert(0,'.')
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.extractor import ORBextractor
B,rows,cols=8,480,640
fr=synth.make_stream(rows,cols,B)
dev=torch.device('cuda',0)
d=torch.from_numpy(fr).to(dev)
ex=ORBextractor(1000)
cap=ex.max_keypoints(rows,cols)
kps=torch.zeros((B,cap,7),device=dev); desc=torch.zeros((B,cap,32),dtype=torch.uint8,device=dev); cnt=torch.zeros((B,2),dtype=torch.int32,device=dev)
ex.extract_batch_device(d,B,rows,cols,cols,rows*cols,kps,desc,cnt,cap,(0,1000),stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
c=cnt.cpu().numpy(); D=desc.cpu().numpy()
a=np.unpackbits(D[3,:c[3,0]],axis=1).astype(np.int32); b=np.unpackbits(D[4,:c[4,0]],axis=1).astype(np.int32)
dist=(a[:,None,:]!=b[None,:,:]).sum(2)
for th in (50,72,100):
    h=(dist<th).sum(1); print(th,"hits/query mean",h.mean(),"max",h.max(),"p90",np.percentile(h,90))
print("dist mean",dist.mean(),"std",dist.std())
