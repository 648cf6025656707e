"""Does computing each convolution's WEIGHT gradient on a side stream (a parallel branch of the captured
graph) shorten the PreResNet-20 training step? The weight gradients are leaves of the backward graph: only
the input gradients are on the critical path."""
import os, sys, tempfile, time
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_fw_'))
import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import models

dev = torch.device('cuda')
aten = torch.ops.aten


class _ForkedConv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, groups, side):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, dilation, groups, b is not None, side)
        return aten.convolution(x, w, b, stride, padding, dilation, False, [0, 0], groups)

    @staticmethod
    def backward(ctx, go):
        x, w = ctx.saved_tensors
        stride, padding, dilation, groups, has_b, side = ctx.cfg
        go = go.contiguous()
        main = torch.cuda.current_stream()
        bs = [w.shape[0]] if has_b else None
        gw = gb = None
        side.wait_stream(main)
        with torch.cuda.stream(side):
            r = aten.convolution_backward(go, x, w, bs, stride, padding, dilation, False, [0, 0], groups, [False, True, has_b])
            gw, gb = r[1], (r[2] if has_b else None)
        gi = None
        if ctx.needs_input_grad[0]:
            gi = aten.convolution_backward(go, x, w, bs, stride, padding, dilation, False, [0, 0], groups, [True, False, False])[0]
        return gi, gw, gb, None, None, None, None, None


class ForkWrw(TorchFunctionMode):
    def __init__(self, side):
        super().__init__()
        self.side = side

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.conv2d:
            x, w = args[0], args[1]
            b = args[2] if len(args) > 2 else kwargs.get('bias')
            stride = args[3] if len(args) > 3 else kwargs.get('stride', 1)
            padding = args[4] if len(args) > 4 else kwargs.get('padding', 0)
            dilation = args[5] if len(args) > 5 else kwargs.get('dilation', 1)
            groups = args[6] if len(args) > 6 else kwargs.get('groups', 1)
            two = lambda v: [v, v] if isinstance(v, int) else list(v)
            if not isinstance(padding, str):
                return _ForkedConv2d.apply(x, w, b, two(stride), two(padding), two(dilation), groups, self.side)
        return func(*args, **kwargs)


def graph_time(fn, steps=200):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


crit = torch.nn.CrossEntropyLoss()
for depth, classes in ((20, 10),):
    for mode in ('plain', 'forked', 'plain', 'forked'):
        torch.manual_seed(0)
        net = models.PreResNet(classes, depth).to(dev)
        x = torch.randn(128, 3, 32, 32, device=dev); y = torch.randint(0, classes, (128,), device=dev)
        params = list(net.parameters())
        side = torch.cuda.Stream()
        net.train()
        grads = {}

        def step():
            if mode == 'forked':
                with ForkWrw(side):
                    loss = crit(net(x), y)
                    loss.backward()
                torch.cuda.current_stream().wait_stream(side)
            else:
                loss = crit(net(x), y)
                loss.backward()
            grads['g'] = [p.grad for p in params]
            for p in params:
                p.grad = None
        ms = graph_time(step)
        chk = float(sum(g.double().sum() for g in grads['g']))
        print(f'PreResNet-{depth} {mode:7s}: {ms:.3f} ms/step   grad checksum {chk:.6f}', flush=True)
