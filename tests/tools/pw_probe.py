"""Timing of the point-wise contraction (LSTM input projections) at the NSVAE shapes.  IDV_PW_CFG selects experimental tiles."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
amd = importlib.import_module("i-dccrn-vae_amd"); ops = amd.ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = 641; Tp = T + 1; dev = "cuda"
line = []
for (M, K) in ((6144, 1280), (3072, 1280), (3072, 768), (1024, 1280)):
    Jp = ops.Planar.jp_for(B, Tp)
    x = torch.randn(K * Jp + 512, device=dev)
    w = torch.randn(M, K, device=dev) * 0.02
    wf, bo = ops.pack_pw(w, torch.zeros(M, device=dev))
    out = torch.empty(B * T * M + 64, device=dev)
    def run():
        ops.pw_gemm(ops.L._P(x.data_ptr() + 1024), K, wf, bo, M, B, Tp, Jp, T, ops.L._P(out.data_ptr()), swap=True, ldo=M)
    try:
        run()
    except TypeError:
        raise
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    line.append(f"M={M} K={K}: {ms:.3f} ms {2*M*K*B*T/ms/1e9:.1f} TF")
print(f"[pw cfg {os.environ.get('IDV_PW_CFG','default')}] " + " | ".join(line), flush=True)
