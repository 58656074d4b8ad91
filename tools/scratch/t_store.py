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
M,N,Kd=802816,512,128
A,B=rnd(M,Kd),rnd(N,Kd)
ob=torch.empty(M,N,device=dev,dtype=torch.bfloat16); of=torch.empty(M,N,device=dev,dtype=torch.float32)
for v in (3,6):
    print(f"v{v} bf16 out: {t(lambda: K.gemm(A,B,out=ob,variant=v)):.3f} ms   (0.82 GB written)")
    print(f"v{v} f32  out: {t(lambda: K.gemm(A,B,out=of,variant=v)):.3f} ms   (1.64 GB written)")
    print(f"v{v} bf16 N=128: {t(lambda: K.gemm(A,B[:128],out=ob[:, :128],variant=v)):.3f} ms  (0.2 GB written, ld 512)")
x=torch.empty(M*N//2,device=dev,dtype=torch.float32)
print(f"torch fill 0.82GB: {t(lambda: x.fill_(1.0)):.3f} ms")
y=torch.empty_like(x)
print(f"torch copy 0.82GB->0.82GB: {t(lambda: y.copy_(x)):.3f} ms")
