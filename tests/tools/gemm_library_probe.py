"""fp32 GEMM throughput of the vendor library (torch.matmul -> hipBLASLt / rocBLAS) at the LSTM input-projection shapes: the yardstick\nfor cgemm_kernel<IDV_PW>.  python tests/tools/gemm_library_probe.py   (GPU box)"""
import torch, time
dev="cuda"
def bench(M,K,N,ta=False,tb=False):
    A=torch.randn((K,M) if ta else (M,K),device=dev); B=torch.randn((N,K) if tb else (K,N),device=dev)
    a=A.t() if ta else A; b=B.t() if tb else B
    for _ in range(3): C=a@b
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): C=a@b
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"M={M} K={K} N={N} ta={ta} tb={tb}: {ms:.3f} ms {2*M*K*N/ms/1e9:.1f} TF",flush=True)
for (M,K,N) in [(6144,1280,20512),(3072,1280,20512),(3072,768,20512),(1024,1280,41088),(1280,6144,20512),(6144,20512,1280)]:
    for ta in (False,True):
        for tb in (False,True):
            bench(M,K,N,ta,tb)
