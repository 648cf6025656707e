import os, sys, json, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from ursabench_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db('ursa_probe_miopen_')
import torch, torch.nn.functional as F
dev = 'cuda'
print('SUGGEST_NHWC', os.environ.get('PYTORCH_MIOPEN_SUGGEST_NHWC'))
# 1x1 conv weight ambiguity: does a contiguous [Cout,Cin,1,1] weight flip the conv to channels_last?
x = torch.randn(128, 64, 8, 8, device=dev)
w1 = torch.randn(256, 64, 1, 1, device=dev)
y = F.conv2d(x, w1)
print('1x1 conv out contiguous:', y.is_contiguous(), 'channels_last:', y.is_contiguous(memory_format=torch.channels_last))
w3 = torch.randn(64, 64, 3, 3, device=dev)
y3 = F.conv2d(x, w3, padding=1)
print('3x3 conv out contiguous:', y3.is_contiguous())
# wrw with NHWC x / dy, dummy channels_last weight; result layout; equality with NCHW wrw
dy = torch.randn_like(y3)
xt, dyt = x.contiguous(memory_format=torch.channels_last), dy.contiguous(memory_format=torch.channels_last)
wd = torch.empty(64, 64, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
gw_t = torch.ops.aten.convolution_backward(dyt, xt, wd, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
gw = torch.ops.aten.convolution_backward(dy, x, w3, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
print('gw_t channels_last:', gw_t.is_contiguous(memory_format=torch.channels_last), 'max rel diff', float((gw_t - gw).abs().max() / gw.abs().max()))
# mixed: NHWC x, dy but NCHW weight?
try:
    gw_m = torch.ops.aten.convolution_backward(dyt, xt, w3, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    print('mixed: gw layout contiguous', gw_m.is_contiguous(), 'cl', gw_m.is_contiguous(memory_format=torch.channels_last))
except Exception as e:
    print('mixed failed', repr(e)[:200])
