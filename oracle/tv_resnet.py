"""
oracle/tv_resnet.py -- TEST INFRASTRUCTURE ONLY.

torchvision.models.resnet{50,101,152} restated as a plain torch nn.Module (v1.5 Bottleneck: stride
on the 3x3 convolution, expansion 4, BatchNorm eps 1e-5, downsample = 1x1 conv(stride) + BN,
blocks (3,4,6,3)/(3,4,23,3)/(3,8,36,3)).  torchvision itself is a third-party dependency that is
neither vendored in the reference nor installed here (pytorch/requirements.txt:8), so this file is
what oracle/reference_shims.py hands to the reference's `models/resnet.py:144-149` in place of
`torchvision.models.resnetNN(weights=...)`; PARITY UNPINNED at that boundary.
"""
import torch as t
from torch import nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class ResNet(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            mods.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*mods)


def resnet50(weights=None):
    return ResNet([3, 4, 6, 3])


def resnet101(weights=None):
    return ResNet([3, 4, 23, 3])


def resnet152(weights=None):
    return ResNet([3, 8, 36, 3])
