import sys, numpy as np, torch, copy, time
sys.path.insert(0,'.')
from video_super_resolution_amd import VSR
from video_super_resolution_amd.weights import fill_module_
m=fill_module_(VSR().eval(),0).cuda()
torch.set_grad_enabled(False)
g=np.load('tests/golden/g4_wrappers.npz')
big=torch.from_numpy(g['flow_frames']).cuda()
for tf32 in (False,):
    flow=m.FlowModule.net(big.permute(3,0,1,2).unsqueeze(0)).cpu().numpy()
    e=np.abs(flow-g['flow']); print('flow err max',e.max(),'range',np.abs(g['flow']).max(),'rel',e.max()/np.abs(g['flow']).max(), 'mean rel', e.mean()/np.abs(g['flow']).max())
pic=m.FlowModule(big[0],big[1]).cpu().numpy(); d=np.abs(pic-g['flow_pic']); print('pic mismatch frac',(d>0).mean(),'max',d.max())
g6=np.load('tests/golden/g6_vsr.npz')
data=torch.from_numpy(g6['data']).cuda()
t=time.time(); out0,_=m(data,None,None,None,train=False); torch.cuda.synchronize(); print('frame time',time.time()-t)
t=time.time(); out0,_=m(data,None,None,None,train=False); torch.cuda.synchronize(); print('frame time 2',time.time()-t)
e=np.abs(out0.cpu().numpy()-g6['out0']); print('e2e err: max',e.max(),'mean',e.mean(),'p50',np.percentile(e,50),'p99',np.percentile(e,99),'p99.9',np.percentile(e,99.9),'range',np.abs(g6['out0']).max())
print(torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32, torch.backends.cudnn.benchmark)
