"""Student / teacher networks of the CIFAR10 configs, as plain nn.Modules.

Shapes follow the reference's ConvolForwardNet and its two specs
(ref: cnn_models/conv_forward_model.py:30-40 specs, :42-163 module): same-padding convs with
ReLU, optional BatchNorm after every conv/linear layer, max-pooling after the listed layers,
ReLU on the output layer too (:160).  Parameter ORDER is kept -- out_layer is registered before
the conv/linear lists (:124 vs :128-132) -- because quantize_first_and_last_layer=False skips
parameters()[0] and [-1] (:237-239).  Only the parameter shapes matter to the quantizer.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

TEACHER_SPEC = dict(conv=[(76, 3), (76, 3), (126, 3), (126, 3), (148, 3), (148, 3), (148, 3), (148, 3)],
                    pool_after=[1, 3, 7], linear=[1200, 1200])           # ~5.3 M parameters
STUDENT_SPEC = dict(conv=[(75, 5), (50, 5), (50, 5), (25, 5)], pool_after=[1, 3], linear=[500])   # ~1.0 M


class ConvNet(nn.Module):
    def __init__(self, conv, pool_after, linear, batch_norm=True, bn_affine=True, classes=10, side=32):
        super().__init__()
        chans = [3] + [c for c, _ in conv]
        flat = chans[-1] * side * side // 4 ** len(pool_after)
        widths = [flat] + list(linear)
        self.out_layer = nn.Linear(widths[-1], classes)                  # registered first, as in the reference
        self.convs = nn.ModuleList(nn.Conv2d(chans[i], chans[i + 1], k, padding=(k - 1) // 2)
                                   for i, (_, k) in enumerate(conv))
        self.linears = nn.ModuleList(nn.Linear(widths[i], widths[i + 1]) for i in range(len(linear)))
        self.norms = nn.ModuleList([nn.BatchNorm2d(c, affine=bn_affine) for c, _ in conv] +
                                   [nn.BatchNorm1d(w, affine=bn_affine) for w in linear]) if batch_norm else None
        self.pool_after = set(pool_after)
        for m in list(self.convs) + list(self.linears) + [self.out_layer]:
            nn.init.xavier_uniform_(m.weight)

    def forward(self, x):
        nconv = len(self.convs)
        for i in range(nconv + len(self.linears)):
            if i == nconv:
                x = x.flatten(1)
            x = F.relu(self.convs[i](x) if i < nconv else self.linears[i - nconv](x))
            if self.norms is not None:
                x = self.norms[i](x)
            if i in self.pool_after:
                x = F.max_pool2d(x, 2)
        return F.relu(self.out_layer(x))


def student():
    return ConvNet(**STUDENT_SPEC)


def teacher():
    return ConvNet(**TEACHER_SPEC)


def kd_loss(student_logits, teacher_logits, labels, temperature=2.0, teacher_weight=0.7):
    """Hinton distillation loss exactly as the reference weighs it
    (ref: cnn_models/help_fun.py:95,124,135-139): 0.7*T^2*KLDiv(log_softmax(zs/T), softmax(zt/T))
    with nn.KLDivLoss()'s default ELEMENT-mean reduction, plus 0.3*cross-entropy."""
    kl = F.kl_div(F.log_softmax(student_logits / temperature, dim=1),
                  F.softmax(teacher_logits / temperature, dim=1), reduction='sum') / student_logits.numel()
    return teacher_weight * temperature ** 2 * kl + (1.0 - teacher_weight) * F.cross_entropy(student_logits, labels)


# ---------------------------------------------------------------------------------------------
class _WideBlock(nn.Module):
    """Pre-activation wide residual block (BN-ReLU-conv3x3, BN-ReLU-conv3x3 [stride], 1x1
    shortcut when the shape changes), biases on, as in the reference's wide_basic
    (ref: cnn_models/wide_resnet.py:27-48).  Dropout is a no-op at rate 0 and omitted."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1, bias=True)
        self.bn2 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, stride=stride, padding=1, bias=True)
        self.shortcut = nn.Conv2d(cin, cout, 1, stride=stride, bias=True) if (stride != 1 or cin != cout) else None

    def forward(self, x):
        out = self.conv1(F.relu(self.bn1(x)))
        out = self.conv2(F.relu(self.bn2(out)))
        return out + (x if self.shortcut is None else self.shortcut(x))


class WideResNet(nn.Module):
    """Wide_ResNet(depth, widen) for 32x32 inputs (ref: cnn_models/wide_resnet.py:50-88):
    conv3x3(3,16), three stages of (depth-4)/6 blocks at widths 16k, 32k, 64k, BN, 8x8 average
    pool, linear.  depth 16 / widen 22 is config 3's student: 60 tensors, 82.7 M parameters."""

    def __init__(self, depth=16, widen=22, classes=10):
        super().__init__()
        assert (depth - 4) % 6 == 0
        n = (depth - 4) // 6
        widths = [16, 16 * widen, 32 * widen, 64 * widen]
        self.conv1 = nn.Conv2d(3, widths[0], 3, padding=1, bias=True)
        blocks, cin = [], widths[0]
        for stage, stride in ((1, 1), (2, 2), (3, 2)):
            for b in range(n):
                blocks.append(_WideBlock(cin, widths[stage], stride if b == 0 else 1))
                cin = widths[stage]
        self.blocks = nn.Sequential(*blocks)
        self.bn1 = nn.BatchNorm2d(widths[3], momentum=0.9)
        self.linear = nn.Linear(widths[3], classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight, gain=2 ** 0.5)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        out = self.blocks(self.conv1(x))
        out = F.avg_pool2d(F.relu(self.bn1(out)), 8)
        return self.linear(out.flatten(1))
