"""One quantized-distillation training step, MI355X-native.

Restates the reference's hot loop (ref: cnn_models/conv_forward_model.py:280-317 with
cnn_models/help_fun.py:60-158) for torch 2.x:

    reference, per step                                  here, per step
    ---------------------------------------------------  -----------------------------------------
    state_dict(); for p: p.data = uniformQuantization()   ONE multi-tensor launch: masters -> shadows
    zero_grad; student fwd; teacher fwd; KD loss; bwd      same (grads land in one flat buffer)
    load_state_dict(saved)  (copy masters back)            nothing: the model computes on the shadows,
                                                           the masters were never overwritten
    [nn.DataParallel reduce/broadcast]                     ONE RCCL all-reduce of the flat gradient
    optimizer.step() on the full-precision weights         SGD (nesterov) on the flat master buffer

The straight-through estimator is the wiring itself: gradients are taken at the quantized point
and applied to the full-precision masters.  mode='per_tensor' keeps the reference's structure
(per-parameter API calls + save/restore) for comparison.
"""
import torch

import quantization
from quantized_distillation_amd import ste
from quantized_distillation_amd.multi_tensor import MultiTensorQuantizer

from . import models
from .flat import FlatLayout, GradSynchronizer, broadcast_from_rank0

STYLES = ('none', 'truncated', 'complicated')       # ref: conv_forward_model.py:213-229


class DistillTrainer(object):
    def __init__(self, student, teacher, device, num_bits=4, bucket_size=256, lr=1e-3, momentum=0.9,
                 weight_decay=2.2e-4, nesterov=True, quantize_first_and_last_layer=True, mode='multi',
                 backprop_quantization_style='none', grad_chunks=1, overlap_allreduce=False, loss_fn=None,
                 clip_norm=None, teacher_stream=False, estimate_quant_grad_every=1, quantize_from_first_step=True):
        """backprop_quantization_style: 'none' (pure STE), 'truncated' (clamp the quantized masters to
        [-1, 1] before quantizing and zero the gradient where |w| > 1) or 'complicated' (bucket-aware
        STE, K7) -- ref: conv_forward_model.py:213-229,240-266; anything else raises as the reference
        does (:228-229).  estimate_quant_grad_every=E: quantize every E-th step only, the steps in
        between run on the full-precision weights (ref: :286,:320-323).  quantize_from_first_step:
        the CNN loop's counter starts at 1, so the first batch is quantized (ref: :198); the seq2seq
        loop's starts at 0, so its first batch is NOT (ref: translation_models/model.py:184,243) --
        pass False for that loop."""
        style = 'none' if backprop_quantization_style is None else str(backprop_quantization_style).lower()
        if style not in STYLES:
            raise ValueError('The specified backprop_quantization_style not recognized')      # ref: :228-229
        if style == 'complicated' and bucket_size is None:
            raise NotImplementedError('Right now the code does not work with bucket_size None.'
                                      ' Not hard to modify though')                           # ref: quant_functions.py:332-334
        if mode not in ('multi', 'per_tensor'):
            raise ValueError("mode must be 'multi' or 'per_tensor'")
        self.device = device
        # optional: the teacher forward (needs neither the quantized weights nor the student) on its own HIP
        # stream beside quantize + student forward, joined before the KD loss.  Measured on the CIFAR step:
        # 456 vs 482 steps/s -- the convolution kernels do not overlap -- so it is off by default.
        self.side = TeacherAhead() if (teacher_stream and torch.device(device).type == 'cuda') else None
        self.student = student.to(device).train()
        self.teacher = teacher.to(device).eval()
        for p in self.teacher.parameters():
            p.requires_grad_(False)
        self.s = 2 ** num_bits                                   # ref: conv_forward_model.py:209-211
        self.bucket_size = bucket_size
        self.mode = mode
        self.style = style
        self.every = max(1, int(estimate_quant_grad_every))
        self._since = 1 if quantize_from_first_step else 0       # step_since_last_grad_quant_estimation
        self._quantized_step = False
        params = list(self.student.parameters())
        self.params = params
        n = len(params)
        self.quantized = [not (not quantize_first_and_last_layer and (i == 0 or i == n - 1)) for i in range(n)]
        layout = FlatLayout([p.shape for p in params])
        self.layout = layout
        self.flat_master = torch.zeros(layout.total, device=device)
        self.flat_grad = torch.zeros(layout.total, device=device)
        self.masters = layout.views(self.flat_master)
        grads = layout.views(self.flat_grad)
        for m, p in zip(self.masters, params):
            m.copy_(p.data)
        # replicas must start identical: only gradients are exchanged afterwards (nn.DataParallel
        # re-broadcasts from GPU 0 every step, ref: resnet34_doublefilters.py:69-70)
        broadcast_from_rank0(self.flat_master, *[b for b in self.student.buffers() if b.is_floating_point()])
        # contiguous runs of quantized parameters in the flat buffers: what the clamp / gradient-mask /
        # un-quantized-step copies work on (first/last parameters excluded, ref: :237-239)
        self.qranges = []
        for i in range(n):
            if self.quantized[i]:
                a, b = layout.offsets[i], layout.end(i)
                if self.qranges and self.qranges[-1][1] == a:
                    self.qranges[-1] = (self.qranges[-1][0], b)
                else:
                    self.qranges.append((a, b))
        if mode == 'multi':
            self.flat_shadow = torch.zeros(layout.total, device=device)
            shadows = layout.views(self.flat_shadow)
            for i, p in enumerate(params):
                # the model computes on the shadow (or straight on the master when not quantized)
                p.data = shadows[i] if self.quantized[i] else self.masters[i]
            qi = [i for i in range(n) if self.quantized[i]]
            self.mt = MultiTensorQuantizer([self.masters[i] for i in qi], self.s, bucket_size,
                                           outputs=[shadows[i] for i in qi])
        else:
            for i, p in enumerate(params):
                p.data = self.masters[i]
        for g, p in zip(grads, params):
            p.grad = g
        self.flat_master.grad = self.flat_grad
        self.opt = torch.optim.SGD([self.flat_master], lr=lr, momentum=momentum, nesterov=nesterov,
                                   weight_decay=weight_decay)
        self.sync = GradSynchronizer(self.flat_grad, chunks=grad_chunks)
        if overlap_allreduce:
            self.sync.attach(params, layout)
        self.loss_fn = loss_fn if loss_fn is not None else cnn_kd_loss_fn
        self.clip_norm = clip_norm

    # ------------------------------------------------------------------ pieces of a step
    def quantize(self):
        """Quantized weights for this step's forward/backward (ref: :235-247); the masters are only
        touched by the 'truncated' clamp, which the reference applies to p.data in place before
        quantizing, i.e. to the full-precision weights that the state_dict still references."""
        if self.style == 'truncated':
            for a, b in self.qranges:                            # ref: :240-241 -- quantized parameters only
                ste.clamp_(self.flat_master[a:b], 1.0)
        if self.mode == 'multi':
            self.mt.quantize(check_pointers=False)
        else:                                                    # the reference's loop shape, :235-247
            for i, p in enumerate(self.params):
                if self.quantized[i]:
                    p.data = quantization.uniformQuantization(self.masters[i], self.s, bucket_size=self.bucket_size)[0]
        self._quantized_step = True

    def skip_quantize(self):
        """A step between two quantized-gradient estimates (estimate_quant_grad_every > 1, or the
        seq2seq loop's first batch): forward/backward on the full-precision weights."""
        if self.mode == 'multi':
            for a, b in self.qranges:
                self.flat_shadow[a:b].copy_(self.flat_master[a:b])
        self._quantized_step = False

    def restore(self):
        if self.mode != 'multi':                                 # ref: :302 load_state_dict
            for i, p in enumerate(self.params):
                p.data = self.masters[i]

    def backward_quant(self):
        """backward_quant_weights_model (ref: :249-266): runs after the weights have been restored and
        the gradient exchanged, before optimizer.step() (:314-318)."""
        if self.style == 'none' or not self._quantized_step:
            return
        if self.style == 'truncated':                            # ref: :263-264 p.grad[p.data.abs() > 1] = 0
            for a, b in self.qranges:
                ste.truncated_ste_(self.flat_grad[a:b], self.flat_master[a:b], 1.0)
        else:                                                    # 'complicated', ref: :265-266 -> quant_functions.py:319-406
            grads = self.layout.views(self.flat_grad)
            for i in range(len(self.params)):
                if self.quantized[i]:
                    ste.ste_bucket_backward(self.masters[i], grads[i], self.bucket_size, self.s, out=grads[i])

    def forward_backward(self, *batch):
        self.flat_grad.zero_()
        loss = self.loss_fn(self.student, self.teacher, *batch, side=self.side)
        loss.backward()
        return loss

    def clip(self):
        """Global gradient-norm clipping on the flat buffer, no host sync (ref: onmt/Optim.py:40-55)."""
        if self.clip_norm is not None:
            norm = self.flat_grad.norm()
            self.flat_grad.mul_((self.clip_norm / (norm + 1e-6)).clamp(max=1.0))

    def step(self, *batch):
        if self._graph_fb is not None:
            return self._step_graph(*batch)
        quantize_now = self._since >= self.every                 # ref: :286 / translation_models/model.py:243
        if quantize_now:
            self.quantize()
        else:
            self.skip_quantize()
        loss = self.forward_backward(*batch)
        if quantize_now:
            self.restore()
        self.sync.sync()
        self.clip()
        self.backward_quant()
        self.opt.step()
        if quantize_now:
            self._since = 0                                      # ref: :320-323
        self._since += 1
        return loss

    # ------------------------------------------------------------------ hipGraph replay of the step
    _graph_fb = None
    _capture_failed = False

    def capture(self, *batch, warmup=3, error_mode='thread_local', settle_s=0.35, _before_capture=None):
        """Capture the launch-bound part of the step in hipGraphs (torch.cuda.CUDAGraph): graph A =
        multi-tensor quantize + student/teacher forward + KD loss + backward, graph B = gradient clipping + the SGD
        update.  The gradient all-reduce stays between the two replays (eager RCCL call), so the
        distributed step is  A.replay(); all_reduce; B.replay().  Only for mode='multi'."""
        self.capture_shapes([batch], warmup=warmup, error_mode=error_mode, settle_s=settle_s, _before_capture=_before_capture)

    def capture_shapes(self, batches, warmup=3, error_mode='thread_local', settle_s=0.35, _before_capture=None):
        """capture() for batches of SEVERAL shapes (the token batches of the seq2seq loop differ in length): one graph A per
        distinct shape, each with its own static input buffers, all sharing one memory pool and the one graph B.

        Safe next to a live RCCL process group (see capture_into / quiesce_collectives below): the capture is thread-local, the
        collectives issued so far have been waited for and retired, and a capture that fails restores the caller's stream and
        leaves this an eager trainer."""
        if self._capture_failed:
            # (measured: capturing again on a trainer whose earlier capture_shapes() died half-way ends in a segmentation fault
            # inside the runtime -- docs/history/profiles/r05_capture_probe.txt; a fresh trainer in the same process captures fine)
            raise RuntimeError('an earlier capture on this trainer failed: discard it and capture on a fresh one')
        assert self.mode == 'multi', 'graph capture needs the persistent-shadow (multi) mode'
        assert self.style == 'none' and self.every == 1 and self._since >= 1, \
            'graph capture covers the plain every-step STE loop only'
        assert self.sync._layout is None or not self.sync.active, \
            'graph capture keeps the all-reduce between the two graphs: not with the overlap hooks of a multi-rank run'
        self.side = None                                 # one captured stream; the graph orders the kernels itself
        shapes = {}
        for b in batches:
            shapes.setdefault(tuple(tuple(t.shape) for t in b), tuple(t.clone() for t in b))
        # The warm-up below runs real steps -- it has to: MIOpen picks its plans, the optimizer creates its momentum buffers --
        # but WITHOUT the gradient exchange, on the sample batches.  Training state must not see them: every rank would update
        # its masters from its own un-reduced gradients and the replicas would stay apart for the rest of the run (only
        # gradients are exchanged afterwards); a single rank would take warmup x #shapes hidden optimizer steps.  So the
        # masters, the model's buffers (batch-norm statistics) and the optimizer state are put back afterwards; the momentum
        # buffers the warm-up created stay allocated (the graph captures their addresses) and are zeroed, which is what a first
        # optimizer step starts from.
        had_state = {id(p): bool(self.opt.state.get(p)) for g in self.opt.param_groups for p in g['params']}
        saved_master = self.flat_master.clone()
        saved_buffers = [b.clone() for b in self.student.buffers()]
        saved_momenta = {id(p): {k: v.clone() for k, v in self.opt.state[p].items() if torch.is_tensor(v)}
                         for g in self.opt.param_groups for p in g['params'] if had_state[id(p)]}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                      # materialises momentum buffers, cudnn/miopen plans
                for sb in shapes.values():
                    self.quantize()
                    self.forward_backward(*sb)
                    self.clip()
                    self.opt.step()
            with torch.no_grad():
                self.flat_master.copy_(saved_master)
                for b, v in zip(self.student.buffers(), saved_buffers):
                    b.copy_(v)
                for g in self.opt.param_groups:
                    for p in g['params']:
                        for k, v in self.opt.state[p].items():
                            if torch.is_tensor(v):
                                v.copy_(saved_momenta[id(p)][k]) if had_state[id(p)] else v.zero_()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        del saved_master, saved_buffers, saved_momenta
        self._capture_failed = True                      # until the whole set of graphs exists
        quiesce_collectives(self.sync, settle_s)
        if _before_capture is not None:                  # test hook (tests/capture_worker.py: collectives in flight on purpose)
            _before_capture()
        graphs, pool, cell = {}, None, {}
        for key, sb in shapes.items():
            g = torch.cuda.CUDAGraph()

            def body_a(sb=sb):
                self.quantize()
                cell['loss'] = self.forward_backward(*sb)
            capture_into(g, body_a, pool=pool, stream=side, error_mode=error_mode)    # the warm-up stream: its BLAS handles / workspaces exist
            pool = g.pool()
            graphs[key] = (sb, g, cell.pop('loss'))
        gb = torch.cuda.CUDAGraph()

        def body_b():
            self.clip()
            self.opt.step()
        capture_into(gb, body_b, pool=pool, stream=side, error_mode=error_mode)
        # only a COMPLETE set of graphs switches the trainer to replay: a capture that raised leaves it an eager trainer
        self._graphs = graphs
        self._graph_fb, self._graph_opt = graphs, gb
        self._capture_failed = False

    def _step_graph(self, *batch):
        entry = self._graphs.get(tuple(tuple(t.shape) for t in batch))
        if entry is None:
            raise ValueError('no graph was captured for a batch of shapes %r' % ([tuple(t.shape) for t in batch],))
        sb, g, loss = entry
        for dst, src in zip(sb, batch):
            dst.copy_(src, non_blocking=True)
        g.replay()
        self.sync.sync()
        self._graph_opt.replay()
        return loss


def quiesce_collectives(sync=None, settle_s=0.35):
    """Before a stream capture in a process that has a c10d process group: wait for every collective this trainer has in
    flight, drain the device, and give the backend's watchdog thread a few of its 100 ms polling periods to retire the finished
    work items.  The watchdog polls the completion EVENT of every work item it still lists (hipEventQuery); with nothing listed
    it has nothing to query while the capture runs.  (Round 4's driver run: that query, from the watchdog thread, hit a
    global-mode capture -> "operation not permitted when stream is capturing" -> std::terminate.)"""
    import time
    import torch.distributed as dist
    if sync is not None:
        for h in getattr(sync, '_handles', ()):
            h.wait()
        sync._handles = []
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized() and settle_s > 0:
        time.sleep(settle_s)


def capture_into(graph, body, pool=None, stream=None, error_mode='thread_local'):
    """body() captured into `graph` (a torch.cuda.CUDAGraph).

    error_mode='thread_local' (hipStreamCaptureModeThreadLocal): only THIS thread's calls are checked against the capture, so
    another thread's event query -- the c10d watchdog's -- neither fails nor invalidates it.  torch's default is 'global'.

    torch.cuda.graph.__exit__ calls capture_end() BEFORE it leaves its stream context; when capture_end() raises, the capture
    stream stays the current stream and everything launched afterwards lands on an invalidated stream ("operation failed due
    to a previous error during capture").  So: whatever happens, the stream that was current before is current again."""
    prev = torch.cuda.current_stream()
    try:
        with torch.cuda.graph(graph, pool=pool, stream=stream, capture_error_mode=error_mode):
            body()
    except BaseException:
        torch.cuda.set_stream(prev)
        raise
    if torch.cuda.current_stream() != prev:              # (not expected: the context restores it on a clean exit)
        torch.cuda.set_stream(prev)


class TeacherAhead(object):
    """Runs the (gradient-free) teacher forward on a second HIP stream so that its small-shape kernels
    overlap the student's; `join` makes the current stream wait for it."""

    def __init__(self):
        self.stream = torch.cuda.Stream()

    def launch(self, teacher, *inputs):
        self.stream.wait_stream(torch.cuda.current_stream())         # inputs were produced on the main stream
        with torch.cuda.stream(self.stream), torch.no_grad():
            return teacher(*inputs)

    def join(self, out):
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.stream)
        out.record_stream(cur)                                       # allocated on the side stream, consumed here
        return out


def _teacher_forward(teacher, side, *inputs):
    """Returns a thunk that yields the teacher output (launched now on the side stream if there is one)."""
    if side is None:
        def late():
            with torch.no_grad():
                return teacher(*inputs)
        return late
    out = side.launch(teacher, *inputs)
    return lambda: side.join(out)


def cnn_kd_loss_fn(student, teacher, images, labels, side=None):
    """Student + teacher forward and the Hinton KD loss (ref: cnn_models/help_fun.py:60-158)."""
    t_out = _teacher_forward(teacher, side, images)
    out = student(images)
    return models.kd_loss(out, t_out(), labels)


def seq2seq_kd_loss_fn(student, teacher, src, tgt, side=None):
    """Teacher-forced student + teacher forward and the word-level KD loss
    (ref: translation_models/help_fun.py:36-84, onmt/Loss.py:97-120).  src, tgt: (len, batch)."""
    tgt_in, tgt_out = tgt[:-1], tgt[1:]
    t_logits = _teacher_forward(teacher, side, src, tgt_in)
    logits = student(src, tgt_in)
    return models.word_kd_loss(logits, t_logits(), tgt_out.flatten(), src.size(1))


def synthetic_token_batch(batch, device, seed=0, v_src=18000, v_tgt=10000, max_len=50):
    """multi30k-shaped synthetic batch: sequence-first int64 (len <= 50, batch), pad id 1."""
    g = torch.Generator().manual_seed(seed)
    s_len = int(torch.randint(20, max_len + 1, (1,), generator=g))
    t_len = int(torch.randint(20, max_len + 1, (1,), generator=g))
    src = torch.randint(2, v_src, (s_len, batch), generator=g)
    tgt = torch.randint(2, v_tgt, (t_len, batch), generator=g)
    tgt[-3:, ::4] = 1                                    # some padded tails
    return src.to(device), tgt.to(device)


def synthetic_batch(batch, device, seed=0, classes=10, side=32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(batch, 3, side, side, generator=g).to(device),
            torch.randint(0, classes, (batch,), generator=g).to(device))
