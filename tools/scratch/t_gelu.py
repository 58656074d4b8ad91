import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pytorch_empirical_mvm_amd import kernels as K
dev="cuda"
def rnd(*s): return torch.randn(*s, device=dev).to(torch.bfloat16)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
for (M,N,Kd) in [(69120,3072,768),(50176,2048,512),(200704,1024,256),(802816,512,128)]:
    A,B=rnd(M,Kd),rnd(N,Kd); bias=torch.zeros(N,device=dev); pre=torch.empty(M,N,device=dev,dtype=torch.bfloat16); u=rnd(M,N)
    Bt=rnd(Kd,N)  # for dgrad shape: dy[M,N] x W^T -> [M,Kd]
    for v in (3,5,6):
        ms=t(lambda: K.gemm(A,B,bias=bias,act=1,out_preact=pre,variant=v)); print(f"fc1+gelu {M}x{N}x{Kd} v{v}: {ms:.3f} ms {2*M*N*Kd/ms/1e9:.0f} TF")
        ms=t(lambda: K.gemm(A,B,bias=bias,variant=v)); print(f"plain    {M}x{N}x{Kd} v{v}: {ms:.3f} ms {2*M*N*Kd/ms/1e9:.0f} TF")
    dy=rnd(M,Kd); W=rnd(N,Kd)   # fc2 dgrad: dh[M,N] = dy[M,Kd] @ W2[Kd... use A=dy [M,Kd], B=W[N,Kd] act=3 aux=u
    for v in (3,5,6):
        ms=t(lambda: K.gemm(dy,W,act=3,aux=u,variant=v)); print(f"dgrad+gelu' {M}x{N}x{Kd} v{v}: {ms:.3f} ms {2*M*N*Kd/ms/1e9:.0f} TF")
