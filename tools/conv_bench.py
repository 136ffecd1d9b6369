import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgcdet_amd import ext
ops = ext.ops()
if os.environ.get("SGC_DIAG_LIB"):      # diagnostic builds (tools/diag): timing only
    from sgcdet_amd._abi import Library
    from sgcdet_amd.tensor_api import TensorOps
    ops = TensorOps(Library(os.environ["SGC_DIAG_LIB"]), "cuda")
if os.environ.get("SGC_HALO_BRICK"):
    ops.lib.call("sgc_set_tuning", b"halo_brick", int(os.environ["SGC_HALO_BRICK"]))
if os.environ.get("SGC_HALO_MIN_M"):
    ops.lib.call("sgc_set_tuning", b"halo_min_m", int(os.environ["SGC_HALO_MIN_M"]))
if os.environ.get("SGC_HALO_MIN"):
    ops.lib.call("sgc_set_tuning", b"halo_min_cout", int(os.environ["SGC_HALO_MIN"]))
layers = [  # name, Cin, Cout, grid, k, s, transposed
 ("down0.conv 256->256 @40x40x16", 256,256,(40,40,16),3,1,False),
 ("out0 256->128 @40x40x16", 256,128,(40,40,16),3,1,False),
 ("down1.conv1 256->512 s2", 256,512,(40,40,16),3,2,False),
 ("down1.conv2 512->512 @20x20x8", 512,512,(20,20,8),3,1,False),
 ("out1 512->128 @20x20x8", 512,128,(20,20,8),3,1,False),
 ("down2.conv1 512->1024 s2", 512,1024,(20,20,8),3,2,False),
 ("down2.conv2 1024->1024 @10x10x4", 1024,1024,(10,10,4),3,1,False),
 ("out2 1024->128 @10x10x4", 1024,128,(10,10,4),3,1,False),
 ("up2.convT 1024->512", 1024,512,(10,10,4),2,2,True),
 ("up1.convT 512->256", 512,256,(20,20,8),2,2,True),
 ("ds1 1x1 s2 256->512", 256,512,(40,40,16),1,2,False),
 ("head 128->32 @40x40x16", 128,32,(40,40,16),3,1,False),
]
for name,Cin,Cout,g,k,s,tr in layers:
    V=g[0]*g[1]*g[2]
    x=torch.randn(V,Cin,device='cuda'); taps=8 if tr else k**3
    wt=torch.randn(taps,Cout,Cin,device='cuda')*0.01
    sc=torch.ones(Cout,device='cuda'); sh=torch.zeros(Cout,device='cuda')
    wh,wl=ops.split_bf16(wt)
    res={}
    for mode in ("f32","bf16x3w4","bf16x3"):
        if mode.startswith("bf16x3"): ops.lib.call("sgc_set_tuning", b"conv_halo", 0 if mode.endswith("w4") else 1)
        f=(lambda: ops.conv3d_cl(x,wt,g,k,s,tr,sc,sh,None,True)) if mode=="f32" else (lambda: ops.conv3d_cl_bf16x3(x,wh,wl,g,k,s,tr,sc,sh,None,True))
        for _ in range(3): y,og=f()
        torch.cuda.synchronize(); t=time.perf_counter()
        n=10
        for _ in range(n): y,og=f()
        torch.cuda.synchronize(); res[mode]=((time.perf_counter()-t)/n, y)
    OV=og[0]*og[1]*og[2]
    fl = 2*Cin*Cout*OV*(1 if tr else taps)
    err=(res["f32"][1]-res["bf16x3"][1]).abs().max().item()/max(1.0,res["f32"][1].abs().max().item())
    print(f"{name:36s} f32 {res['f32'][0]*1e6:8.1f} us {fl/res['f32'][0]/1e12:6.1f} TF | bf16x3 per-tap {res['bf16x3w4'][0]*1e6:8.1f} us | halo {res['bf16x3'][0]*1e6:8.1f} us {fl/res['bf16x3'][0]/1e12:6.1f} TF-eq | rel diff {err:.1e}")
