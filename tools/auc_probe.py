import sys, torch
sys.path.insert(0, '.')
from manner_amd import hip
g = torch.Generator(device='cuda').manual_seed(1)
n = 2604035
s = torch.randn(n, device='cuda', generator=g) * 3 + 775
l = (torch.rand(n, device='cuda', generator=g) < 0.07).float()
for _ in range(5):
    a = hip.auc(s, l)
torch.cuda.synchronize()
print(float(a))
