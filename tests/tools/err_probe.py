import importlib, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_backward as T
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops
pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
errs = []
orig = T.check
def check(name, got, want, tol=T.GTOL, atol=0.0):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    e = float((got - want).norm() / (want.norm() + 1e-30))
    errs.append((name, e))
T.check = check
for case in [(True, 256, 128, 5, 37, 2, 256, True), (False, 32, 64, 17, 40, 2, 0, True), (True, 32, 1, 33, 41, 2, 32, True), (True, 16, 1, 17, 20, 2, 16, True), (True, 32, 1, 129, 161, 2, 32, True), (False, 1, 32, 257, 161, 2, 0, True)]:
    errs.clear()
    T.test_conv_block_grads(ops, pm, None, *case)
    print(case, " ".join(f"{n.split('.')[-2] if '.' in n else n}.{n.split('.')[-1]}={e:.1e}" for n, e in errs))
