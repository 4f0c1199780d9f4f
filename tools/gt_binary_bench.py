import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gopairingbasedcryptography_amd import bn254
g1, g2 = bn254.generators()
n = 1 << 18
rng = np.random.default_rng(1)
k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 31] &= 0x1f
P = torch.from_numpy(bn254.g1_scalar_mul(g1, k.reshape(-1))).cuda()
Q = torch.from_numpy(np.repeat(g2[None], n, 0)).cuda()
gt = bn254.pair_batch(P, Q)
for name, fn in (("gt_div", lambda: bn254.gt_div(gt, gt)), ("gt_inverse", lambda: bn254.gt_inverse(gt)), ("gt_mul", lambda: bn254.gt_mul(gt, gt))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print("%s %.1f M/s" % (name, 5 * n / (time.perf_counter() - t0) / 1e6))
