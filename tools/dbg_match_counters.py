import sys, numpy as np, torch, ctypes as C
sys.path.insert(0,'.')
from orbslam3_cpp_fork_amd import synth, _lib
from orbslam3_cpp_fork_amd.extractor import ORBextractor
from orbslam3_cpp_fork_amd.matcher import ORBmatcher
B,rows,cols=64,480,640
fr=synth.make_stream(rows,cols,B)
dev=torch.device('cuda',0)
ts=torch.cuda.Stream(); torch.cuda.set_stream(ts); st=ts.cuda_stream
d=torch.from_numpy(fr).to(dev)
ex=ORBextractor(1000); mt=ORBmatcher(0.7,True)
cap=ex.max_keypoints(rows,cols)
kps=torch.zeros((B,cap,7),device=dev); desc=torch.zeros((B,cap,32),dtype=torch.uint8,device=dev); cnt=torch.zeros((B,2),dtype=torch.int32,device=dev)
pairs=torch.from_numpy(np.stack([(np.arange(B)-1)%B,np.arange(B)],1).astype(np.int32)).to(dev)
m=torch.zeros((B,cap),dtype=torch.int32,device=dev); nm=torch.zeros(B,dtype=torch.int32,device=dev)
ex.extract_batch_device(d,B,rows,cols,cols,rows*cols,kps,desc,cnt,cap,(0,1000),stream=st)
torch.cuda.synchronize()
L=_lib.lib()
L.orbm_debug_counters(mt._h,1,None)
mt.match_pairs_device(desc,kps,cnt,cap,pairs,B,m,nm,stream=st)
out=np.zeros(3,np.int32); L.orbm_debug_counters(mt._h,0,out.ctypes.data_as(C.c_void_p))
print("rounds,rescans,chunks per pair:", out/B, "matches", nm.float().mean().item())
