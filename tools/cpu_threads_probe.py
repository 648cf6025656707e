import time, torch, os, sys
sys.path.insert(0, os.getcwd())
from ursabench_amd import models
net = models.PreResNet(10, 20)
x = torch.randn(128,3,32,32); y = torch.randint(0,10,(128,))
crit = torch.nn.CrossEntropyLoss()
print('cpu_count', os.cpu_count())
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    for _ in range(2):
        net.zero_grad(); crit(net(x), y).backward()
    t0=time.perf_counter()
    for _ in range(5):
        net.zero_grad(); crit(net(x), y).backward()
    print(nt, 'threads', (time.perf_counter()-t0)/5*1e3, 'ms/step')
