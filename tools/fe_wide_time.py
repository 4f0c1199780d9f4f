import time, numpy as np, os, sys, torch
sys.path.insert(0, os.getcwd())
from gopairingbasedcryptography_amd import bn254
g1, g2 = bn254.generators()
dP = torch.from_numpy(g1[None].copy()).cuda(); dQ = torch.from_numpy(g2[None].copy()).cuda()
f = bn254.miller_loop(dP, dQ); torch.cuda.synchronize()
for _ in range(3): e = bn254.final_exp(f); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    e = bn254.final_exp(f); torch.cuda.synchronize()
print(os.environ.get("GPBC_LIB_PATH", "in-tree"), "final_exp alone %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
