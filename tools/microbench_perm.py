import sys, torch
sys.path.insert(0, '/root/repo')
from nnx_ppo_amd import random as rnd
k = rnd.key(3, torch.device('cuda:0'))
for n in (4096, 8192):
    for _ in range(3): rnd.permutations(k, 4, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): rnd.permutations(k, 4, n)
    e1.record(); torch.cuda.synchronize()
    print(n, 'us per call', e0.elapsed_time(e1) / 20 * 1e3)
