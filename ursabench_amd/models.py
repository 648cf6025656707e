"""The networks the benchmark configurations name (BASELINE.json `configs`), as plain nn.Modules.

Linear layers stay stock PyTorch-ROCm (rocBLAS), and so do convolutions outside a gradient-recording step. Two things these
modules do differently from the reference's on a HIP device: the convolutions of a training step (`fused_conv.Conv2d`: K8 / K9
forward and input gradient, K7 weight gradient - every convolution of the BasicBlock ResNets, instead of MIOpen's launches) and
`relu(bn(x))`: every BatchNorm of the
pre-activation networks is followed by a ReLU, and that pair runs as the K6 launches of
`fused_bn.bn_relu` (2 forward + 2 backward per layer instead of 5-9 MIOpen / ATen launches: the
BatchNorm + ReLU share of a PreResNet-20 training step was 31 % of its kernel time). Host tensors
take the stock ops. The samplers accept any nn.Module (URSABench/inference/sghmc.py:66).

state_dict keys, parameter order and initialisation distributions follow the reference
classes so a reference state_dict loads unchanged (pinned by tests/golden/model_keys.json):
  PreResNet   URSABench/models/preresnet.py:90-151   (depth 20 -> BasicBlock, 272,282 params)
  WideResNet  URSABench/models/wideresnet.py:78-120  (28-10, 100 classes -> 36,546,980 params)
  MLP         URSABench/models/mlp.py:8-23
LeNet5 is not in the reference: classic 2-conv / 3-fc LeNet-5 on 1x28x28.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused_block, fused_bottleneck
from .fused_bn import add_bn_relu, bn_relu
from .fused_conv import Conv2d

__all__ = ['LeNet5', 'MLP', 'MLP_dropout', 'PreResNet', 'PreResNet_dropout', 'WideResNet', 'MLP200MNIST',
           'MLP200MNIST_dropout', 'LeNet5MNIST', 'PreResNet8', 'PreResNet20', 'PreResNet164', 'WideResNet28x10']


class LeNet5(nn.Module):
    def __init__(self, num_classes=10):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 6, 5, padding=2)
        self.conv2 = nn.Conv2d(6, 16, 5)
        self.fc1 = nn.Linear(16 * 5 * 5, 120)
        self.fc2 = nn.Linear(120, 84)
        self.fc3 = nn.Linear(84, num_classes)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.conv1(x)), 2)
        x = F.max_pool2d(F.relu(self.conv2(x)), 2)
        x = x.flatten(1)
        return self.fc3(F.relu(self.fc2(F.relu(self.fc1(x)))))


class MLP(nn.Module):
    def __init__(self, hidden_size, input_dim, num_classes):
        super().__init__()
        self.input_dim, self.hidden_size, self.num_classes = input_dim, hidden_size, num_classes
        self.fc1 = nn.Linear(input_dim, hidden_size)
        self.fc2 = nn.Linear(hidden_size, hidden_size)
        self.fc3 = nn.Linear(hidden_size, num_classes)

    def forward(self, x):
        h = F.relu(self.fc1(x.view(-1, self.input_dim)))
        return self.fc3(F.relu(self.fc2(h)))


class MLP_dropout(nn.Module):
    """URSABench/models/mlp.py:25-41 — dropout through F.dropout with its default training=True, i.e. the
    masks stay ON in eval mode: that is what makes T forwards of ONE model an MC-dropout ensemble
    (inference/vi_dropout.py returns the same live model `num_samples` times)."""

    def __init__(self, hidden_size, input_dim, num_classes, dropout=0.2):
        super().__init__()
        self.input_dim, self.hidden_size, self.num_classes = input_dim, hidden_size, num_classes
        self.fc1 = nn.Linear(input_dim, hidden_size)
        self.fc2 = nn.Linear(hidden_size, hidden_size)
        self.fc3 = nn.Linear(hidden_size, num_classes)
        self.dropout = dropout

    def forward(self, x):
        x = self.fc1(x.view(-1, self.input_dim))
        x = self.fc2(F.relu(F.dropout(x, p=self.dropout)))
        return self.fc3(F.relu(F.dropout(x, p=self.dropout)))


# ---- pre-activation ResNet --------------------------------------------------------------
class _PreActBasic(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(cin)
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.downsample = downsample

    def forward(self, x):
        """`x`: a tensor or the pending sum (a, b) the previous block returned; returns the pending sum
        (conv output, shortcut): the `out += residual` of preresnet.py:49-52 is done by whoever consumes it - the next
        block's bn1 or the network's final bn (`fused_bn.add_bn_relu`: same values, one launch less each way)."""
        x, h = add_bn_relu(self.bn1, x)
        y = self.conv1(h)
        y = self.conv2(bn_relu(self.bn2, y))
        return y, (x if self.downsample is None else self.downsample(x))


class _PreActBottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(cin)
        self.conv1 = Conv2d(cin, planes, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        # pending sums in and out, as in _PreActBasic. K13 (fused_bottleneck): the 1x1 layers apply relu(bn(.)) while they stage
        # their input - large activations with gradients recorded (the HMC configuration); K6 + K12 otherwise
        ab = x if isinstance(x, tuple) else (x,)
        if fused_bottleneck.eligible(self.bn1, self.conv1, *ab):
            x, y = fused_bottleneck.bn_relu_conv1x1(self.bn1, self.conv1, x)
        else:
            x, h = add_bn_relu(self.bn1, x)
            y = self.conv1(h)
        y = self.conv2(bn_relu(self.bn2, y))
        if fused_bottleneck.eligible(self.bn3, self.conv3, y):
            y = fused_bottleneck.bn_relu_conv1x1(self.bn3, self.conv3, y)[1]
        else:
            y = self.conv3(bn_relu(self.bn3, y))
        return y, (x if self.downsample is None else self.downsample(x))


def _pool8(x, pool=None):
    """The final AvgPool2d(8) of the CIFAR networks (preresnet.py:110, wideresnet.py:116). On the 8x8 map they all
    end with, that is the mean over the map: ATen's reduction (5 us) instead of its avg_pool2d kernel (31 us forward on
    [128, 64, 8, 8], profiles/r03_bench_kernel_stats.csv) — same value to fp32 rounding. Other map sizes, and host
    tensors (the CPU replays of the reference's runs stay op-for-op the reference's): the pooling op."""
    if x.is_cuda and x.shape[-2:] == (8, 8):
        return x.mean((2, 3))
    return (F.avg_pool2d(x, 8) if pool is None else pool(x)).flatten(1)


class PreResNet(nn.Module):
    """depth = 6n+2 (BasicBlock, depth < 44) or 9n+2 (Bottleneck)."""

    def __init__(self, num_classes=10, depth=110):
        super().__init__()
        self.num_classes, self.depth = num_classes, depth     # read back by vi_dropout.change_to_dropout_model
        if depth >= 44:
            if (depth - 2) % 9:
                raise AssertionError('depth should be 9n+2')
            reps, block = (depth - 2) // 9, _PreActBottleneck
        else:
            if (depth - 2) % 6:
                raise AssertionError('depth should be 6n+2')
            reps, block = (depth - 2) // 6, _PreActBasic
        self.inplanes = 16
        self.conv1 = Conv2d(3, 16, 3, padding=1, bias=False)
        self.layer1 = self._stage(block, 16, reps, 1)
        self.layer2 = self._stage(block, 32, reps, 2)
        self.layer3 = self._stage(block, 64, reps, 2)
        self.bn = nn.BatchNorm2d(64 * block.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.avgpool = nn.AvgPool2d(8)
        self.fc = nn.Linear(64 * block.expansion, num_classes)
        for m in self.modules():                       # preresnet.py:114-120
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _stage(self, block, planes, reps, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False))
        blocks = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, reps)]
        return nn.Sequential(*blocks)

    def _trunk(self, x):
        """relu(bn(layer3(layer2(layer1(conv1(x)))))). A training step of a BasicBlock network on a HIP device takes the K10
        launches (`fused_block`: one launch per bn -> relu -> conv unit); everything else the K6 / K8 launches or the stock ops."""
        if self.depth < 44 and fused_block.eligible(self, x):
            return fused_block.trunk(self, x)
        if self.depth < 44 and fused_block.eval_eligible(self, x):       # an ensemble member's forward: no gradient, running statistics
            return bn_relu(self.bn, fused_block.eval_trunk(self, x))
        x = self.layer3(self.layer2(self.layer1(self.conv1(x))))     # the last block's pending sum
        return add_bn_relu(self.bn, x)[1]

    def forward(self, x):
        return self.fc(_pool8(self._trunk(x), self.avgpool))

    def forward_loss(self, x, target, crit):
        """crit(self(x), target) for a training step whose loss is a plain mean cross entropy, with the head - the last BatchNorm, ReLU,
        pooling, classifier, loss and their gradients - as three launches (`fused_block.trunk_loss`, K11) instead of ~13; None when
        that does not apply (the caller then evaluates crit(self(x), target) itself). The chain engine asks."""
        if type(self) is PreResNet and self.depth < 44 and fused_block.head_eligible(self, x, target, crit):
            return fused_block.trunk_loss(self, x, target, crit)
        return None


class PreResNet_dropout(PreResNet):
    """Not in the reference (it ships MLP_, ResNet_ and WideResNet_dropout only): the benchmark's PreResNet with
    always-on dropout before the classifier, placed like ResNet_dropout's (imagenet_resnet.py:141), so that
    MCdropout can run on BASELINE configs[1]'s network."""

    def __init__(self, num_classes=10, depth=110, dropout=0.2):
        super().__init__(num_classes, depth)
        self.dropout = dropout

    def forward(self, x):
        return self.fc(F.dropout(_pool8(self._trunk(x), self.avgpool), p=self.dropout))


# ---- wide ResNet -------------------------------------------------------------------------
class _WideBlock(nn.Module):
    def __init__(self, cin, planes, dropout_rate, stride=1):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(cin)
        self.conv1 = Conv2d(cin, planes, 3, padding=1, bias=True)
        self.dropout = nn.Dropout(p=dropout_rate)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride, 1, bias=True)
        self.shortcut = nn.Sequential()
        if stride != 1 or cin != planes:
            self.shortcut = nn.Sequential(Conv2d(cin, planes, 1, stride, bias=True))

    def forward(self, x):
        x, h = add_bn_relu(self.bn1, x)                 # pending sums in and out, as in _PreActBasic
        y = self.dropout(self.conv1(h))
        y = self.conv2(bn_relu(self.bn2, y))
        return y, self.shortcut(x)


class WideResNet(nn.Module):
    def __init__(self, num_classes=10, depth=28, widen_factor=10, dropout_rate=0.):
        super().__init__()
        if (depth - 4) % 6:
            raise AssertionError('Wide-resnet depth should be 6n+4')
        reps, k = (depth - 4) // 6, widen_factor
        self.in_planes = 16
        self.conv1 = Conv2d(3, 16, 3, padding=1, bias=True)
        self.layer1 = self._stage(16 * k, reps, dropout_rate, 1)
        self.layer2 = self._stage(32 * k, reps, dropout_rate, 2)
        self.layer3 = self._stage(64 * k, reps, dropout_rate, 2)
        self.bn1 = nn.BatchNorm2d(64 * k, momentum=0.9)
        self.linear = nn.Linear(64 * k, num_classes)

    def _stage(self, planes, reps, dropout_rate, stride):
        blocks = []
        for s in [stride] + [1] * (reps - 1):
            blocks.append(_WideBlock(self.in_planes, planes, dropout_rate, s))
            self.in_planes = planes
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.layer3(self.layer2(self.layer1(self.conv1(x))))
        return self.linear(_pool8(add_bn_relu(self.bn1, x)[1]))


# ---- config classes: `.base/.args/.kwargs` like URSABench/models (preresnet.py:154-169) ----
class _Cfg:
    args = list()
    kwargs = dict()
    transform_train = None      # the benchmark feeds synthetic device-resident tensors
    transform_test = None


class MLP200MNIST(_Cfg):
    base = MLP
    kwargs = {'hidden_size': 200, 'input_dim': 784}


class MLP200MNIST_dropout(_Cfg):
    base = MLP_dropout
    kwargs = {'hidden_size': 200, 'input_dim': 784, 'dropout': 0.2}


class LeNet5MNIST(_Cfg):
    base = LeNet5


class PreResNet8(_Cfg):
    base = PreResNet
    kwargs = {'depth': 8}


class PreResNet20(_Cfg):
    base = PreResNet
    kwargs = {'depth': 20}


class PreResNet164(_Cfg):
    base = PreResNet
    kwargs = {'depth': 164}


class WideResNet28x10(_Cfg):
    base = WideResNet
    kwargs = {'depth': 28, 'widen_factor': 10}
