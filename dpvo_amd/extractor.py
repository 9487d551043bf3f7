"""Feature encoders of the Patchifier (reference dpvo/extractor.py:6-55,200-264).

The weight containers: plain torch.nn modules whose names / parameter names follow the reference so that `dpvo.pth` state dicts load
strictly.  The tracker does NOT run their forward: dpvo_amd/encoders.py packs their weights once and runs both towers through the HIP
implicit-GEMM kernels of csrc/encoder.hip.  The torch forward below (MIOpen convolutions through PyTorch-ROCm) is
what tests/test_gpu_encoders.py and the reference-generated goldens compare those kernels with."""
import torch
import torch.nn as nn

DIM = 32


class ResidualBlock(nn.Module):
    def __init__(self, in_planes, planes, norm_fn='group', stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.relu = nn.ReLU(inplace=True)

        def make_norm():
            if norm_fn == 'group':
                return nn.GroupNorm(num_groups=planes // 8, num_channels=planes)
            if norm_fn == 'batch':
                return nn.BatchNorm2d(planes)
            if norm_fn == 'instance':
                return nn.InstanceNorm2d(planes)
            if norm_fn == 'none':
                return nn.Sequential()
            raise ValueError(norm_fn)

        self.norm1 = make_norm()
        self.norm2 = make_norm()
        if not stride == 1:
            self.norm3 = make_norm()
        if stride == 1:
            self.downsample = None
        else:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)

    def forward(self, x):
        y = x
        y = self.relu(self.norm1(self.conv1(y)))
        y = self.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class BasicEncoder4(nn.Module):
    def __init__(self, output_dim=128, norm_fn='batch', dropout=0.0, multidim=False):
        super().__init__()
        self.norm_fn = norm_fn
        self.multidim = multidim
        if norm_fn == 'group':
            self.norm1 = nn.GroupNorm(num_groups=8, num_channels=DIM)
        elif norm_fn == 'batch':
            self.norm1 = nn.BatchNorm2d(DIM)
        elif norm_fn == 'instance':
            self.norm1 = nn.InstanceNorm2d(DIM)
        elif norm_fn == 'none':
            self.norm1 = nn.Sequential()
        self.conv1 = nn.Conv2d(3, DIM, kernel_size=7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.in_planes = DIM
        self.layer1 = self._make_layer(DIM, stride=1)
        self.layer2 = self._make_layer(2 * DIM, stride=2)
        self.conv2 = nn.Conv2d(2 * DIM, output_dim, kernel_size=1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _make_layer(self, dim, stride=1):
        layer1 = ResidualBlock(self.in_planes, dim, self.norm_fn, stride=stride)
        layer2 = ResidualBlock(dim, dim, self.norm_fn, stride=1)
        self.in_planes = dim
        return nn.Sequential(layer1, layer2)

    def forward(self, x):
        b, n, c1, h1, w1 = x.shape
        x = x.view(b * n, c1, h1, w1)
        x = self.conv1(x)
        x = self.norm1(x)
        x = self.relu1(x)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.conv2(x)
        _, c2, h2, w2 = x.shape
        return x.view(b, n, c2, h2, w2)
