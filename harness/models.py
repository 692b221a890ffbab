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


# ---------------------------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + (x if self.down is None else self.down(x)))


class ResNetK(nn.Module):
    """ImageNet ResNet of BasicBlocks with every width multiplied by k
    (ref: cnn_models/resnet_kfilters.py:82-140).  layers=(2,2,2,2), k=1.5 is config 4's student
    (62 tensors, 25.9 M parameters); layers=(3,4,6,3), k=1 stands in for the ResNet-34 teacher."""

    def __init__(self, layers=(2, 2, 2, 2), k=1.0, classes=1000):
        super().__init__()
        w = [int(64 * k), int(128 * k), int(256 * k), int(512 * k)]
        self.conv1 = nn.Conv2d(3, w[0], 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(w[0])
        blocks, cin = [], w[0]
        for stage, (width, count) in enumerate(zip(w, layers)):
            for b in range(count):
                blocks.append(_BasicBlock(cin, width, 2 if (b == 0 and stage > 0) else 1))
                cin = width
        self.blocks = nn.Sequential(*blocks)
        self.fc = nn.Linear(cin, classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                nn.init.normal_(m.weight, 0.0, (2.0 / fan) ** 0.5)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, stride=2, padding=1)
        x = self.blocks(x)
        return self.fc(F.adaptive_avg_pool2d(x, 1).flatten(1))


# ---------------------------------------------------------------------------------------------
class Seq2SeqLSTM(nn.Module):
    """2-layer LSTM encoder-decoder with input feeding and 'general' global attention, the default
    model of the reference's translation configs (ref: onmt/standard_options.py:19-38,
    onmt/Models.py, onmt/modules/StackedRNN.py:10-18, GlobalAttention.py:50-58,
    ModelConstructor.py:150-152): 22 parameter tensors -- 2 embeddings (V,500), encoder nn.LSTM
    (8), decoder stacked cells (2000,1000)/(2000,500) + biases (8), attention linear_in (500,500)
    and linear_out (500,1000), generator (V_tgt,500)+(V_tgt).  Sequence-first tensors (len, batch).
    """

    def __init__(self, v_src=18000, v_tgt=10000, emb=500, hidden=500, layers=2):
        super().__init__()
        self.src_emb = nn.Embedding(v_src, emb, padding_idx=1)
        self.tgt_emb = nn.Embedding(v_tgt, emb, padding_idx=1)
        self.encoder = nn.LSTM(emb, hidden, num_layers=layers)
        self.dec_cells = nn.ModuleList([nn.LSTMCell(emb + hidden if i == 0 else hidden, hidden) for i in range(layers)])
        self.attn_in = nn.Linear(hidden, hidden, bias=False)
        self.attn_out = nn.Linear(2 * hidden, hidden, bias=False)
        self.generator = nn.Linear(hidden, v_tgt)
        self.hidden = hidden

    def forward(self, src, tgt_in):
        """src: (S, B) int64, tgt_in: (T, B) int64 -> logits (T*B, V_tgt)."""
        memory, (h, c) = self.encoder(self.src_emb(src))                 # (S, B, H)
        mem_t = memory.transpose(0, 1)                                   # (B, S, H)
        keys = self.attn_in(mem_t)                                       # 'general' score: h^T W m
        hs, cs = list(h.unbind(0)), list(c.unbind(0))
        feed = memory.new_zeros(src.size(1), self.hidden)
        emb = self.tgt_emb(tgt_in)
        outs = []
        for t in range(tgt_in.size(0)):
            x = torch.cat([emb[t], feed], dim=1)                         # input feeding
            for i, cell in enumerate(self.dec_cells):
                hs[i], cs[i] = cell(x, (hs[i], cs[i]))
                x = hs[i]
            score = torch.bmm(keys, x.unsqueeze(2)).squeeze(2)            # (B, S)
            ctx = torch.bmm(F.softmax(score, dim=1).unsqueeze(1), mem_t).squeeze(1)
            feed = torch.tanh(self.attn_out(torch.cat([ctx, x], dim=1)))
            outs.append(feed)
        return self.generator(torch.stack(outs).flatten(0, 1))


def word_kd_loss(student_logits, teacher_logits, target, batch, pad=1, teacher_weight=0.7):
    """Word-level distillation loss of the reference's NMT path (ref: onmt/Loss.py:40-55,97-120):
    0.3 * NLL + 0.7 * KL(teacher || student) at T = 1, padding positions masked, summed over
    tokens and divided by the batch size."""
    mask = (target != pad).float()
    logp = F.log_softmax(student_logits, dim=1)
    nll = -(logp.gather(1, target.clamp(min=0).unsqueeze(1)).squeeze(1) * mask).sum()
    pt = F.softmax(teacher_logits, dim=1)
    kl = ((pt * (torch.log(pt.clamp_min(1e-30)) - logp)).sum(dim=1) * mask).sum()
    return ((1.0 - teacher_weight) * nll + teacher_weight * kl) / batch
