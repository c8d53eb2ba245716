import sys, torch
sys.path.insert(0, ".")
from tris_amd import ops
torch.manual_seed(0)
for (M,N,K) in [(4096,4096,4096),(2400,768,3072),(307200,64,256)]:
    A=torch.randn(M,K,device="cuda"); B=torch.randn(N,K,device="cuda"); C0=torch.empty(M,N,device="cuda"); C1=torch.empty_like(C0)
    ops.set_gemm_mode("f32"); ops.gemm(A,B,C0,M,N,K,K,K,N,False,True,use_ws=False)
    ops.set_gemm_mode("x3"); ops.gemm(A,B,C1,M,N,K,K,K,N,False,True,use_ws=False)
    ref=(A[:256].double()@B.double().t())
    e0=(C0[:256].double()-ref).abs().max().item(); e1=(C1[:256].double()-ref).abs().max().item()
    print(M,N,K,"f32 err vs fp64",e0,"x3 err vs fp64",e1,"scale",ref.abs().max().item())
    for mode in ("f32","x3"):
        ops.set_gemm_mode(mode)
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        for _ in range(3): ops.gemm(A,B,C1,M,N,K,K,K,N,False,True,use_ws=False)
        a.record()
        for _ in range(10): ops.gemm(A,B,C1,M,N,K,K,K,N,False,True,use_ws=False)
        b.record(); torch.cuda.synchronize()
        ms=a.elapsed_time(b)/10
        print("   ",mode,f"{ms*1e3:.1f} us {2.0*M*N*K/(ms*1e-3)/1e12:.1f} TF/s")
