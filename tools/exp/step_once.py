import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import models
dev = torch.device('cuda')
torch.manual_seed(0)
net = models.PreResNet(10, 20).to(dev)
x = torch.randn(128, 3, 32, 32, device=dev); y = torch.randint(0, 10, (128,), device=dev)
crit = torch.nn.CrossEntropyLoss()
net.train()
for i in range(30):
    if i == 10:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    net.zero_grad(); crit(net(x), y).backward()
torch.cuda.synchronize()
print('eager ms/step', (time.perf_counter() - t0) / 20 * 1e3)
