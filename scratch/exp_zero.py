import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3
hip=_native.Context()
lib=C.CDLL('/root/repo/scratch/exp_conv.so')
P=C.c_void_p
lib.exp_conv.argtypes=[P,P,P,P,P,C.c_int,C.c_int,C.c_int,C.c_int,C.c_int,C.c_float,C.c_int,P,C.c_int]
def bench(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
B,c,H,W=16,48,256,3072
w=torch.randn(c,c,3,3)/np.sqrt(9*c); b=torch.randn(c,device='cuda')
pk,un=pack_conv3x3(w.numpy()); wp=torch.from_numpy(pk.view(np.int16)).cuda(); wz=torch.zeros_like(wp)
for name,x,ww in (("random", torch.randn(B,c,H,W,device='cuda'), wp), ("zero weights", torch.randn(B,c,H,W,device='cuda'), wz)):
    out=torch.empty_like(x)
    for mode,mn in [(0,'full'),(7,'MFMA only'),(6,'LDS reads + MFMA')]:
        t=bench(lambda: lib.exp_conv(hip._h, x.data_ptr(), ww.data_ptr(), b.data_ptr(), out.data_ptr(), B,c,c,H,W, un, 1, None, mode))
        print(f"{name:14s} {mn:18s} {t:.3f} ms")
