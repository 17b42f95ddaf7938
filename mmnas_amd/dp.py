"""Data-parallel gradient exchange for the bilevel NAS loop: one process per GPU,
torch.distributed over RCCL/xGMI (backend "nccl" on ROCm), replacing the reference's
DistributedDataParallel wrapper (search_vqa.py:210, train_vqa.py:236).

What is different from DDP, and why:
  * every parameter gradient lives in ONE flat fp32 device buffer (p.grad are views), so a
    collective moves contiguous memory and no bucket copies exist;
  * fixed architecture (Net_Full): the buffer is cut into a few large buckets in backward order; a
    bucket's all-reduce is issued on a side stream as soon as its last gradient has been
    accumulated, overlapping the rest of backward (xGMI is point-to-point, 7 links per GPU: few,
    large messages, not DDP's 25 MB default);
  * supernet weight step: only the stem/head and the SAMPLED candidates' gradients are exchanged
    (gathered into a staging buffer by one HIP kernel, one all-reduce, scattered back) -- the
    reference all-reduces all 148 MB including the ~2/3 that are zeros because of its
    `0 * sum(p.sum())` trick (search_vqa.py:285-288);
  * supernet arch step: only the [n_nodes, 4] alpha_gate gradient block is exchanged (<= 120 floats);
  * the sampled architecture itself is kept identical on all ranks by the seeded CPU sampler in
    model/mixed.py (optionally verified with `check_same_architecture`).
Gradients are AVERAGED over ranks, as DDP does; the loss keeps reduction='sum' per rank.
"""
import ctypes as C

import torch
import torch.distributed as dist


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _has_avg(group=None):
    """ReduceOp.AVG exists on RCCL only; gloo (CPU tests, and the two-ranks-on-one-GPU test) sums and scales."""
    return dist.get_backend(group) == 'nccl'


def _align(n, a=64):
    return (n + a - 1) // a * a


class _Sink:
    """Handle a parameter carries (`p._mmnas_sink`) while its gradient lives in a flat buffer: the HIP
    backward kernels accumulate straight into `view` (mmnas_amd.ops._grad_bufs) and call `done()`."""
    __slots__ = ('view', 'index', 'callback')

    def __init__(self, view, index, callback):
        self.view, self.index, self.callback = view, index, callback

    def done(self):
        if self.callback is not None:
            self.callback(self.index)


class FlatGrads:
    """Flat gradient storage: `views[i]` is the gradient view of `params[i]` inside `flat`."""

    def __init__(self, params):
        self.params = [p for p in params]
        dev = self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += _align(p.numel())
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        self.index = {id(p): i for i, p in enumerate(self.params)}

    def attach(self, which=None):
        """Point p.grad at its view (all parameters, or the given subset; others get grad=None)."""
        if which is None:
            for p, v in zip(self.params, self.views):
                p.grad = v
            return
        keep = {id(p) for p in which}
        for p, v in zip(self.params, self.views):
            p.grad = v if id(p) in keep else None

    def zero(self):
        self.flat.zero_()

    def enable_sinks(self, callback=None):
        """Let the operators' backward kernels write parameter gradients directly into the views
        (no per-operator zero-fill, no autograd accumulate kernel).  `callback(i)` fires when the
        gradient of params[i] has been enqueued."""
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            p._mmnas_sink = _Sink(v, i, callback)

    def disable_sinks(self):
        for p in self.params:
            if hasattr(p, '_mmnas_sink'):
                del p._mmnas_sink


class GradReducer:
    """Bucketed, backward-overlapped all-reduce of a fixed parameter set (Net_Full training)."""

    def __init__(self, params, bucket_mb=64.0, group=None):
        self.fg = FlatGrads(params)
        self.group = group
        self.world = _world()
        self.is_cuda = self.fg.flat.is_cuda
        self.fg.attach()
        # buckets over the flat buffer, filled in REVERSE parameter order (~ backward order)
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets = []  # (lo, hi, [param indices])
        hi = self.fg.total
        cur = []
        lo = hi
        for i in reversed(range(len(self.fg.params))):
            lo = self.fg.offsets[i]
            cur.append(i)
            if hi - lo >= cap:
                self.buckets.append((lo, hi, cur))
                hi, cur = lo, []
        if cur:
            self.buckets.append((lo, hi, cur))
        self.bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._seen = [True] * len(self.fg.params)   # armed by begin_step()
        self._works = []
        self._launched = [False] * len(self.buckets)
        self.comm_stream = torch.cuda.Stream() if (self.is_cuda and self.world > 1) else None
        if self.world > 1:
            for i, p in enumerate(self.fg.params):
                # The hook is the ONLY arrival signal.  Autograd runs a parameter's AccumulateGrad node -- and
                # this hook -- once per backward, after every use of the parameter has run its backward, also
                # when the operator wrote the gradient through a sink and handed autograd None.  (Counting the
                # sinks' done() as well made every sink parameter arrive twice, so a bucket could be reduced
                # before its last gradients were enqueued.)
                p.register_post_accumulate_grad_hook(self._make_hook(i))
        if self.is_cuda:
            self.fg.enable_sinks(None)   # HIP backward kernels add straight into the flat buffer

    def _arrived(self, i):
        if self._seen[i]:
            return
        self._seen[i] = True
        b = self.bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _make_hook(self, i):
        def hook(_p):
            self._arrived(i)
        return hook

    def _launch(self, b):
        if self._launched[b]:
            return
        self._launched[b] = True
        lo, hi, _ = self.buckets[b]
        chunk = self.fg.flat[lo:hi]
        avg = _has_avg(self.group)
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        if self.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._works.append((dist.all_reduce(chunk, op=op, group=self.group, async_op=True), chunk, not avg))
        else:
            self._works.append((dist.all_reduce(chunk, op=op, group=self.group, async_op=True), chunk, not avg))

    def begin_step(self):
        """Call before forward: zero the gradient buffer and arm the buckets."""
        self.fg.zero()
        self.fg.attach()
        if self.world == 1:
            return
        self._pending = [len(idxs) for (_, _, idxs) in self.buckets]
        self._seen = [False] * len(self.fg.params)
        self._launched = [False] * len(self.buckets)
        self._works = []

    def finish(self):
        """Call after backward: flush buckets whose gradients never arrived, wait for all collectives."""
        if self.world == 1:
            return
        for b in range(len(self.buckets)):
            self._launch(b)
        if self.is_cuda:
            with torch.cuda.stream(self.comm_stream):   # (the scaling of a summed bucket stays on the comm stream)
                for w, chunk, need_div in self._works:
                    w.wait()
                    if need_div:
                        chunk.mul_(1.0 / self.world)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w, chunk, need_div in self._works:
                w.wait()
                if need_div:
                    chunk.mul_(1.0 / self.world)
        self._works = []


class SupernetReducer:
    """Gradient exchange for Net_Search: stem/head + sampled candidates on weight steps, the
    alpha_gate block on arch steps."""

    def __init__(self, net, group=None):
        self.net = net
        self.group = group
        self.world = _world()
        mops = net.redundant_modules
        cand = set()
        for m in mops:
            for p in m.candidate_ops.parameters():
                cand.add(id(p))
        self.shared = [p for p in net.net_parameters() if id(p) not in cand]       # stem + head
        self.per_op = [[list(op.parameters()) if op is not None else [] for op in m.candidate_ops] for m in mops]
        ordered = list(self.shared)
        for node in self.per_op:
            for ps in node:
                ordered += ps
        self.fg = FlatGrads(ordered)
        self.is_cuda = self.fg.flat.is_cuda
        self.staging = None
        self._active = None
        if self.is_cuda:
            self.fg.enable_sinks(None)   # HIP backward kernels add straight into the flat buffer

    def _segments(self, params):
        """Merge the flat ranges of `params` into maximal contiguous (offset, n) runs."""
        rng = sorted((self.fg.offsets[self.fg.index[id(p)]], _align(p.numel())) for p in params)
        out = []
        for o, n in rng:
            if out and out[-1][0] + out[-1][1] == o:
                out[-1][1] += n
            else:
                out.append([o, n])
        return out

    def begin_weight_step(self):
        """After reset_binary_gates(): zero the buffer, give gradient views to the stem/head and the
        sampled candidates only (unsampled candidates keep grad=None, mixed.py:160-163)."""
        active = list(self.shared)
        for m, node in zip(self.net.redundant_modules, self.per_op):
            for i in m.active_index:
                active += node[i]
        self.fg.zero()
        self.fg.attach(active)
        self._active = active

    def finish_weight_step(self):
        if self.world == 1:
            return
        segs = self._segments(self._active)
        total = sum(n for _, n in segs)
        if self.staging is None or self.staging.numel() < total:
            self.staging = torch.empty(total, dtype=torch.float32, device=self.fg.flat.device)
        stg = self.staging[:total]
        self._pack(segs, stg, 0)
        if _has_avg(self.group):
            dist.all_reduce(stg, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(stg, op=dist.ReduceOp.SUM, group=self.group)
            stg.mul_(1.0 / self.world)
        self._pack(segs, stg, 1)

    def _pack(self, segs, stg, direction):
        if self.is_cuda:
            from . import _lib as L
            arr = (L.Segment * len(segs))()
            base = self.fg.flat.data_ptr()
            off = 0
            for k, (o, n) in enumerate(segs):
                arr[k].ptr, arr[k].offset, arr[k].n = base + 4 * o, off, n
                off += n
            # the table rides in the kernel arguments: no host->device copy, no stream synchronisation per step
            L.check(L.lib().mmnas_pack_segments_host(arr, len(segs), L.fptr(stg), 1.0, direction, L.stream()))
        else:  # CPU tensors (gloo tests): host plumbing only
            off = 0
            for o, n in segs:
                if direction == 0:
                    stg[off:off + n].copy_(self.fg.flat[o:o + n])
                else:
                    self.fg.flat[o:o + n].copy_(stg[off:off + n])
                off += n

    def reduce_alpha_gate_grads(self):
        """Arch step: average dL/dgate over ranks (the only gradient the arch step uses, mixed.py:172)."""
        if self.world == 1:
            return
        mops = self.net.redundant_modules
        width = max(m.n_choices for m in mops)
        dev = mops[0].alpha_gate.device
        g = torch.zeros(len(mops), width, device=dev)
        for i, m in enumerate(mops):
            if m.alpha_gate.grad is not None:
                g[i, :m.n_choices] = m.alpha_gate.grad
        if _has_avg(self.group):
            dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            g.mul_(1.0 / self.world)
        for i, m in enumerate(mops):
            if m.alpha_gate.grad is not None:
                m.alpha_gate.grad.copy_(g[i, :m.n_choices])


def check_same_architecture(net, group=None):
    """Debug aid: assert every rank sampled the same operators (they share the sampler seed)."""
    if _world() == 1:
        return True
    idx = torch.tensor([m.active_index[0] for m in net.redundant_modules], dtype=torch.int64)
    dev = next(net.parameters()).device
    idx = idx.to(dev)
    lo, hi = idx.clone(), idx.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


def broadcast_parameters(module, src=0, group=None):
    """Initial parameter sync from rank 0 (what DDP's constructor does, search_vqa.py:210)."""
    if _world() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
