import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Camera, Engine, RoadWidthParams
H,W,B=512,1024,4
eng=Engine(H,W,B,"resnet50")
wf=Wt.make_fcn8s_weights(1,decoder_std=0.05); wm=Wt.make_monodepth_weights("resnet50",2)
eng.load_weights(L.SD_NET_FCN8S,wf); eng.load_weights(L.SD_NET_MONODEPTH,wm)
rng=np.random.default_rng(1000)
base=rng.integers(0,256,(B,H//8,W//8,3),dtype=np.uint8)
fr=np.repeat(np.repeat(base,8,axis=1),8,axis=2)
fr=(fr.astype(np.int16)+rng.integers(-16,17,fr.shape,dtype=np.int16)).clip(0,255).astype(np.uint8)
frames=torch.from_numpy(fr).cuda()
cams=[Camera(W/2,H/2,1000.0,1.0,float(W))]*B
out=eng.process_batch(frames,cams,RoadWidthParams())
pp=out["disp_pp"].cpu().numpy()
print("disp_pp percentiles", np.percentile(pp,[0,1,10,50,90,99,100]))
print("per-row mean range", pp[0].mean(1).min(), pp[0].mean(1).max(), "per-col", pp[0].mean(0).min(), pp[0].mean(0).max())
rec=Engine.records(out["records"])
for r in rec: print({k:(r[k].tolist() if hasattr(r[k],'tolist') else r[k]) for k in rec.dtype.names})
print("road frac", out["seg"]["road"].float().mean().item(), "fence", out["seg"]["fence"].float().mean().item())
# cpu timing parts
from oracle import nets, pipeline
for nt in (256, 64, 32, 16):
    torch.set_num_threads(nt)
    t=time.time(); lg=nets.fcn8s_forward(fr[:1],wf); t1=time.time()-t
    f=fr[0].astype(np.float32)/255; pair=np.stack((f,np.fliplr(f)),0)
    t=time.time(); d=nets.monodepth_forward(pair,wm,"resnet50"); t2=time.time()-t
    print("threads",nt,"fcn",round(t1,2),"mono",round(t2,2), flush=True)
_,road,fence,_=nets.softmax_masks(lg[0])
t=time.time(); pipeline.frame_tail(d[...,0].astype(np.float32),road,fence,fr[0],dict(cx=W/2,cy=H/2,f=1000.0,b=1.0,disp_mult=float(W))); print("tail",time.time()-t)
