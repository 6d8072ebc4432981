import sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native
from audio_cut_amd.separation.conv_pack import pack_conv3x3
hip=_native.Context()
lib=C.CDLL('/root/repo/scratch/exp_ws.so')
P=C.c_void_p
lib.exp_conv_ws.argtypes=[P,P,P,P,P,C.c_int,C.c_int,C.c_int,C.c_int,C.c_int,C.c_float,C.c_int,P]
lib.ac_last_error.restype=C.c_char_p
def bench(fn,n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
shapes=[(2,48,16,64),(16,48,256,3072),(16,96,128,1536),(16,144,64,768),(16,192,32,384),(16,288,8,96)]
if len(sys.argv)>1: shapes=shapes[:1]
for (B,c,H,W) in shapes:
    x=torch.randn(B,c,H,W,device='cuda'); w=torch.randn(c,c,3,3)/np.sqrt(9*c); b=torch.randn(c,device='cuda')
    pk,un=pack_conv3x3(w.numpy()); wp=torch.from_numpy(pk.view(np.int16)).cuda()
    out=torch.empty_like(x); out2=torch.zeros_like(x)
    hip.conv3x3_f16x3(x, wp, b, c, un, relu=True, out=out)
    rc=lib.exp_conv_ws(hip._h, x.data_ptr(), wp.data_ptr(), b.data_ptr(), out2.data_ptr(), B,c,c,H,W, un, 1, None)
    torch.cuda.synchronize()
    if rc: print("rc",rc, lib.ac_last_error()); break
    eq=torch.equal(out,out2)
    t1=bench(lambda: hip.conv3x3_f16x3(x, wp, b, c, un, relu=True, out=out))
    t2=bench(lambda: lib.exp_conv_ws(hip._h, x.data_ptr(), wp.data_ptr(), b.data_ptr(), out2.data_ptr(), B,c,c,H,W, un, 1, None))
    print(f"B{B} c{c} {H}x{W}: main {t1:.3f} ms | warp-specialised persistent {t2:.3f} ms | equal {eq} maxdiff {(out-out2).abs().max().item():.2e}")
