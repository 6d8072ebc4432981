import sys, time, subprocess, threading, numpy as np, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3
hip=_native.Context()
B,c,H,W=16,96,128,1536
x=torch.randn(B,c,H,W,device='cuda'); w=torch.randn(c,c,3,3)/np.sqrt(9*c); b=torch.randn(c,device='cuda')
pk,un=pack_conv3x3(w.numpy()); wp=torch.from_numpy(pk.view(np.int16)).cuda(); out=torch.empty_like(x)
xz=torch.zeros_like(x)
samples=[]
stop=False
def sampler():
    while not stop:
        try:
            o=subprocess.run(["rocm-smi","--showpower","--showclocks","--showtemp","--json"],capture_output=True,text=True,timeout=5).stdout
            samples.append((time.time(), o))
        except Exception as e:
            samples.append((time.time(), "ERR "+str(e)))
        time.sleep(0.3)
th=threading.Thread(target=sampler); th.start()
time.sleep(1.0)
for name,inp in (("random",x),("zeros",xz)):
    torch.cuda.synchronize(); t0=time.time(); n=0
    while time.time()-t0<4.0:
        for _ in range(50): hip.conv3x3_f16x3(inp, wp, b, c, un, relu=True, out=out)
        torch.cuda.synchronize(); n+=50
    dt=time.time()-t0
    print(name, "avg ms", dt/n*1e3, "window", t0, t0+dt, flush=True)
    time.sleep(1.0)
stop=True; th.join()
import json
for t,o in samples:
    try:
        d=json.loads(o); k=list(d.keys())[0]; c0=d[k]
        keys=[kk for kk in c0 if any(s in kk.lower() for s in ("power","sclk","mclk","temperature (sensor edge)","junction"))]
        print(round(t,1), {kk:c0[kk] for kk in keys})
    except Exception:
        print(round(t,1), o[:200].replace("\n"," "))
