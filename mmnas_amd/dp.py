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
import os

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
    __slots__ = ('view', 'index', 'callback', 'ready_cb', 'owner', 'fg', 'uses', 'gen')

    def __init__(self, view, index, callback, ready_cb=None, owner=None, fg=None):
        self.view, self.index, self.callback, self.ready_cb, self.owner = view, index, callback, ready_cb, owner
        # live section nodes (ops.BackboneFn / HeadFn forwards whose backward has not run yet) that add into this view
        # in the current step; `gen` ties the count to the step it was made in (FlatGrads.zero() starts a new one)
        self.fg, self.uses, self.gen = fg, 0, -1

    def live(self):
        if self.fg is not None and self.gen != self.fg.gen:
            self.gen, self.uses = self.fg.gen, 0
        return self.uses

    def acquire(self):
        """A section forward that will run a backward has taken this parameter (train_itm.py:380-391: three forwards
        share every weight before the one backward)."""
        self.live()
        self.uses += 1

    def release(self):
        """That section's backward has enqueued its share.  True when it was the last live one: the parameter's WHOLE
        gradient of this step is now enqueued."""
        if self.live() > 0:
            self.uses -= 1
        return self.uses == 0

    def done(self):
        """One operator has enqueued its contribution (a parameter shared by several operators gets several)."""
        if self.callback is not None:
            self.callback(self.index)

    def ready(self):
        """The parameter's WHOLE gradient of this backward pass has been enqueued (sent by ops.BackboneFn, whose
        parameters are not autograd inputs and therefore never reach a post-accumulate hook)."""
        if self.ready_cb is not None:
            self.ready_cb(self.index)


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
        # dirty[i]: the view of params[i] may hold non-zero data (it has been attached as p.grad, or a gradient was
        # copied into it) since the last zero().  FlatAdam's dense mode relies on "not dirty => all zeros".
        self.dirty = [False] * len(self.params)
        self.gen = 0   # step generation: bumped by zero(); _Sink.live() keys its live-node count on it

    def attach(self, which=None):
        """Point p.grad at its view (all parameters, or the given subset; others get grad=None)."""
        if which is None:
            for p, v in zip(self.params, self.views):
                p.grad = v
            self.dirty = [True] * len(self.params)
            return
        keep = {id(p) for p in which}
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if id(p) in keep:
                p.grad = v
                self.dirty[i] = True
            else:
                p.grad = None

    def zero(self):
        self.flat.zero_()
        self.dirty = [False] * len(self.params)
        self.gen += 1

    def adopt(self, i):
        """Bring a gradient that was produced OUTSIDE the flat buffer into it: the driver called `net.zero_grad()` /
        set `p.grad = None` after the views were attached (the reference loop does, search_vqa.py:290), so backward
        allocated a fresh tensor.  The view takes its value (the buffer was zeroed when the step began, and nothing
        else wrote into this view since: sinks only write while `p.grad is view`) and becomes p.grad again."""
        p, v = self.params[i], self.views[i]
        g = p.grad
        if g is None or g is v:
            return False
        if g.data_ptr() != v.data_ptr():
            v.copy_(g)
        p.grad = v
        self.dirty[i] = True
        return True

    def adopt_strays(self, which=None):
        n = 0
        for i in (range(len(self.params)) if which is None else which):
            n += bool(self.adopt(i))
        return n

    def enable_sinks(self, callback=None, ready_cb=None, owner=None):
        """Let the operators' backward kernels write parameter gradients directly into the views
        (no per-operator zero-fill, no autograd accumulate kernel).  `callback(i)` fires when one operator has
        enqueued its share of params[i]'s gradient, `ready_cb(i)` when the whole gradient has been (see _Sink)."""
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            p._mmnas_sink = _Sink(v, i, callback, ready_cb, owner, self)

    def disable_sinks(self):
        for p in self.params:
            if hasattr(p, '_mmnas_sink'):
                del p._mmnas_sink


def _chain_marks(red, op_params, bucket_of, buckets):
    """Events for ops.BackboneFn.backward, which issues the whole backbone's backward in ONE native call (operators in
    reverse index order): for every bucket whose still-missing gradients all belong to chain operators, an event behind
    the operator that issues the bucket's LAST gradient.  The chain records it there (mmnas_chain.marks); the bucket's
    all-reduce then waits on it instead of on the end of the call, i.e. it overlaps the backward of the operators in
    front.  Returns one entry per operator (None = no mark), or None when there is nothing to mark."""
    if not (red.comm and red.is_cuda):
        return None
    index = red.fg.index
    last_op = {}                      # param index -> the operator (smallest index = issued last) that completes it
    for i, ps in enumerate(op_params):
        for p in ps:
            j = index.get(id(p))
            if j is not None and (j not in last_op or i < last_op[j]):
                last_op[j] = i
    marks = [None] * len(op_params)
    any_mark = False
    for b in buckets:
        if red._launched[b]:
            continue
        pend = red._pending_params(b)
        if not pend or any(j not in last_op for j in pend):
            continue                  # (a stem parameter is still missing: the bucket completes later, through its hook)
        i = min(last_op[j] for j in pend)
        if marks[i] is None:
            marks[i] = torch.cuda.Event()
            marks[i].record(torch.cuda.current_stream())   # (creates the handle; the chain records it again in place)
        red._mark_ev[b] = marks[i]
        any_mark = True
    red.marks_made = getattr(red, 'marks_made', 0) + sum(m is not None for m in marks)
    return marks if any_mark else None


# MMNAS_DP_TAIL_MAIN=0 restores round 5's end of step (everything on the communication stream, one scatter per bucket)
_TAIL_ON_MAIN = os.environ.get('MMNAS_DP_TAIL_MAIN', '1') != '0'


def _row_sparse_index(fg, comm, is_cuda):
    """Index (in fg.params) of the parameter whose gradient is exchanged as rows, or None.  The nets tag their word
    embedding (`weight._mmnas_row_sparse`): its gradient is the ~900 rows of the batch's tokens inside a 24 MB table, it
    completes LAST in backward (nothing is left to overlap a dense all-reduce with) and is half of everything the
    d = 256 supernet exchanges per step.  Only the first parameter of the flat buffer is taken (a bucket then simply
    starts behind it)."""
    if not (comm and is_cuda and fg.params) or os.environ.get('MMNAS_DP_ROWS', '1') == '0':   # (0: dense, for A/B runs)
        return None
    return 0 if getattr(fg.params[0], '_mmnas_row_sparse', False) else None


class RowExchange:
    """Data-parallel exchange of a row-sparse gradient (nn.Embedding.weight): instead of all-reducing the dense [V, E]
    table (DDP, search_vqa.py:292), every rank all-gathers the batch's token indices and output-gradient rows (~1 MB per
    rank) and adds ALL ranks' rows, scaled by 1 / world, into its own (zeroed) gradient view with a fixed summation order
    (mmnas_embedding_bwd_det) -- so the ranks' results are bitwise equal, as an all-reduce's are.  Runs on the reducer's
    communication stream from inside ops.EmbeddingFn.backward.  A gradient that reaches the parameter any other way
    (a dense autograd gradient) is caught at finish() and all-reduced densely.  Only a backward between the reducer's
    begin and finish takes this path (`active`): anywhere else -- the arch step, whose weight gradients nobody reads and
    which never joins the communication stream -- the embedding's gradient is the local scatter-add on the main stream."""

    def __init__(self, red, index):
        self.red, self.i = red, index
        self.done = False
        self.active = False       # True between the reducer's begin and finish: only then does a backward take the row path
        self._keep = []
        self._idx = {}
        self._next_key = 0

    def begin(self):
        self.done = False
        self.active = True
        self._keep = []
        self._idx = {}
        self._next_key = 0

    def gather_indices(self, idx):
        """Forward (ops.EmbeddingFn.forward): the ranks' token indices are exchanged while the step computes, so that
        only the gradient rows are left for the end of backward.  Returns a key for exchange()."""
        red = self.red
        idx_l = idx.reshape(-1).contiguous()
        world = dist.get_world_size(red.group)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        cs = red.comm_stream
        with torch.cuda.stream(cs):
            cs.wait_event(ev)
            ip = [torch.empty_like(idx_l) for _ in range(world)]
            dist.all_gather(ip, idx_l, group=red.group)
            idx_all = torch.cat(ip) if world > 1 else ip[0]
        idx_l.record_stream(cs)
        key = self._next_key          # never reused within a step: exchange() pops entries while forwards may still add
        self._next_key += 1
        self._idx[key] = (idx_l, ip, idx_all)
        return key

    def exchange(self, idx, dy, key=None):
        from . import _lib as L
        red = self.red
        view = red.fg.views[self.i]
        V, E = view.shape
        if key is None or key not in self._idx:
            key = self.gather_indices(idx)
        idx_l, ip, idx_all = self._idx.pop(key)
        dy_l = dy.reshape(-1, E).contiguous()
        world = dist.get_world_size(red.group)
        if _TAIL_ON_MAIN:
            # the embedding's backward is the LAST node of backward: its rows are exchanged on the backward's own stream (the
            # all-gather of the indices, issued in forward, is joined first) -- two stream hand-overs less at the end of the step
            cs = torch.cuda.current_stream()
            cs.wait_stream(red.comm_stream)
        else:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            cs = red.comm_stream
        with torch.cuda.stream(cs):
            if not _TAIL_ON_MAIN:
                cs.wait_event(ev)
            gp = [torch.empty_like(dy_l) for _ in range(world)]
            dist.all_gather(gp, dy_l, group=red.group)
            dy_all = torch.cat(gp) if world > 1 else gp[0]
            ws = torch.empty(idx_all.numel() * E, dtype=torch.float32, device=dy_all.device)
            L.check(L.lib().mmnas_embedding_bwd_det(L.ptr(idx_all), L.fptr(dy_all), L.fptr(view), L.fptr(ws), idx_all.numel(), E, V,
                                                    1.0 / world, cs.cuda_stream))
        dy_l.record_stream(cs)
        # (several calls per step -- the ITM triplet step embeds three captions -- simply accumulate, in call order on
        #  every rank.)  Held until finish(): the communication stream still reads them
        self._keep.append((idx_l, dy_l, idx_all, dy_all, ip, gp, ws))
        red.fg.dirty[self.i] = True
        self.done = True

    def finish(self):
        """Called by the reducer's finish, before it joins the communication stream.  Dense fallback when the gradient
        did not come through exchange() this step."""
        red = self.red
        if not self.done:
            red.fg.adopt(self.i)
            view = red.fg.views[self.i]
            if red.is_cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                with torch.cuda.stream(red.comm_stream):
                    red.comm_stream.wait_event(ev)
                    _all_reduce_avg(view, red.group, red.world)
            else:
                _all_reduce_avg(view, red.group, red.world)
        self._keep = []
        self._idx = {}
        self.active = False


def _inline_collectives(group, is_cuda):
    """RCCL collectives issued as SYNCHRONOUS ops from inside `with torch.cuda.stream(comm_stream)`: c10d then enqueues
    the collective on that stream itself, in order behind the pack kernel and in front of the scatter -- no internal
    communication stream, no event pair per bucket (each cross-stream hop is ~10 us of idle queue at the end of a step,
    where nothing is left to overlap it).  The host does not block (a wait on a CUDA work object only orders streams).
    MMNAS_DP_INLINE=0 restores async_op=True + work.wait()."""
    if not is_cuda or os.environ.get('MMNAS_DP_INLINE', '1') == '0':
        return False
    try:
        return dist.get_backend(group) == 'nccl'
    except Exception:   # noqa: BLE001
        return False


def _all_reduce_avg(t, group, world):
    if _has_avg(group):
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.mul_(1.0 / world)


class GradReducer:
    """Bucketed, backward-overlapped all-reduce of a fixed parameter set (Net_Full training)."""

    def __init__(self, params, bucket_mb=64.0, group=None, force_collectives=False):
        """force_collectives: issue the collectives also in a one-rank group (tests of the RCCL path on a one-GPU box)."""
        self.fg = FlatGrads(params)
        self.group = group
        self.world = _world()
        self.comm = self.world > 1 or (force_collectives and dist.is_initialized())
        self.is_cuda = self.fg.flat.is_cuda
        self.fg.attach()
        ri = _row_sparse_index(self.fg, self.comm, self.is_cuda)
        self.row_exchange = RowExchange(self, ri) if ri is not None else None
        # buckets over the flat buffer, filled in REVERSE parameter order (~ backward order)
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets = []  # (lo, hi, [param indices])
        hi = self.fg.total
        cur = []
        lo = hi
        for i in reversed(range(len(self.fg.params))):
            if i == ri:
                continue          # (index 0: the last bucket starts behind it)
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
        self.comm_stream = torch.cuda.Stream() if (self.is_cuda and self.comm) else None
        self.inline = self.comm and _inline_collectives(group, self.is_cuda)
        if self.comm:
            for i, p in enumerate(self.fg.params):
                # The hook is the ONLY arrival signal.  Autograd runs a parameter's AccumulateGrad node -- and
                # this hook -- once per backward, after every use of the parameter has run its backward, also
                # when the operator wrote the gradient through a sink and handed autograd None.  (Counting the
                # sinks' done() as well made every sink parameter arrive twice, so a bucket could be reduced
                # before its last gradients were enqueued.)
                p.register_post_accumulate_grad_hook(self._make_hook(i))
        if self.is_cuda:
            # HIP backward kernels add straight into the flat buffer; the backbone chain reports its parameters itself
            self.fg.enable_sinks(None, self._arrived if self.comm else None, self if self.comm else None)
        self._mark_ev = {}

    def _pending_params(self, b):
        return [i for i in self.buckets[b][2] if not self._seen[i]]

    def chain_marks(self, op_params):
        return _chain_marks(self, op_params, self.bucket_of, range(len(self.buckets)))

    def _arrived(self, i):
        if self._seen[i]:
            return
        self._seen[i] = True
        if i not in self.bucket_of:      # the row-exchanged parameter: RowExchange.exchange() / .finish() move it
            return
        self.fg.adopt(i)   # (a gradient autograd accumulated outside the flat buffer: see FlatGrads.adopt)
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
            from . import ops
            ev = self._mark_ev.pop(b, None)      # recorded inside the backbone chain, behind the bucket's last operator
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
            ops.side_stream_barrier(self.comm_stream)   # weight gradients the backbone chain put on its side stream
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if self.inline:       # on the communication stream itself, in stream order
                    dist.all_reduce(chunk, op=op, group=self.group)
                    self._works.append((None, chunk, not avg))
                else:
                    self._works.append((dist.all_reduce(chunk, op=op, group=self.group, async_op=True), chunk, not avg))
        else:
            self._works.append((dist.all_reduce(chunk, op=op, group=self.group, async_op=True), chunk, not avg))

    def begin_step(self):
        """Call before forward: zero the gradient buffer and arm the buckets."""
        self.fg.zero()
        self.fg.attach()
        if not self.comm:
            return
        self._mark_ev = {}
        self._pending = [len(idxs) for (_, _, idxs) in self.buckets]
        self._seen = [False] * len(self.fg.params)
        self._launched = [False] * len(self.buckets)
        self._works = []
        if self.row_exchange is not None:
            self.row_exchange.begin()

    def finish(self):
        """Call after backward: flush buckets whose gradients never arrived, wait for all collectives."""
        if not self.comm:
            self.fg.adopt_strays()
            return
        if self.row_exchange is not None:
            self.row_exchange.finish()
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self.fg.adopt_strays(self.buckets[b][2])
            self._launch(b)
        if self.is_cuda:
            with torch.cuda.stream(self.comm_stream):   # (the scaling of a summed bucket stays on the comm stream)
                for w, chunk, need_div in self._works:
                    if w is not None:
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
    """Gradient exchange for Net_Search: stem/head + sampled candidates on weight steps, the alpha_gate block on arch
    steps.

    Layout: the flat buffer holds [stem | head | rel-stem | node 0 cand 0 | node 0 cand 1 | ... ], every candidate one
    contiguous run, so a step's exchange set is a STATIC segment per shared run plus one segment per sampled
    candidate -- nothing is sorted or merged per step.  The segments are dealt to `n_buckets` buckets in backward order
    (head and the last decoder nodes first, the encoder and the stem last); a bucket is gathered into its own staging
    slice by one kernel and all-reduced on a side stream as soon as autograd has accumulated its last gradient, while
    backward continues with the earlier nodes (search_vqa.py:292 behind DDP's bucketing, but over ~1/3 of the bytes)."""

    def __init__(self, net, group=None, n_buckets=3, force_collectives=False, attach_all=False):
        """attach_all: every parameter keeps its gradient view attached for good (unsampled candidates then show a ZERO
        gradient instead of None -- what the reference loop's `0 * sum(p.sum())` terms produce, search_vqa.py:285-288);
        saves re-pointing ~900 `.grad` attributes per step.  Only the exchange set still follows the sample."""
        self.net = net
        self.attach_all = attach_all
        self.group = group
        self.world = _world()
        self.comm = self.world > 1 or (force_collectives and dist.is_initialized())
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
        self._active = None
        fg = self.fg
        ri = _row_sparse_index(fg, self.comm, self.is_cuda)
        self.row_exchange = RowExchange(self, ri) if ri is not None else None

        def span(params):   # contiguous by construction
            if not params:
                return None
            i0, i1 = fg.index[id(params[0])], fg.index[id(params[-1])]
            return (fg.offsets[i0], fg.offsets[i1] + _align(params[-1].numel()) - fg.offsets[i0])

        # shared parameters: head = used after the backbone (its gradients arrive FIRST in backward)
        head_ids = set()
        for name in ('attflat_x', 'attflat_y', 'attfc_y', 'proj_norm', 'proj', 'proj_scores', 'proj_reg'):
            mod = getattr(net, name, None)
            if mod is not None:
                head_ids.update(id(p) for p in mod.parameters())
        runs = []   # maximal runs of consecutive shared parameters of the same kind: (is_head, [param indices])
        for p in self.shared:
            i = fg.index[id(p)]
            if i == ri:
                continue          # exchanged as rows (RowExchange), not as part of a bucket
            h = id(p) in head_ids
            if runs and runs[-1][0] == h and runs[-1][1][-1] == i - 1:
                runs[-1][1].append(i)
            else:
                runs.append((h, [i]))
        # units in backward order: head runs, nodes from the last to the first, stem runs
        n = len(mops)
        node_size = [max((sum(_align(p.numel()) for p in ps) for ps in node), default=0) for node in self.per_op]
        units = [('run', r) for r in runs if r[0]] + [('node', k) for k in reversed(range(n))] + [('run', r) for r in runs if not r[0]]
        sizes = [sum(_align(fg.params[i].numel()) for i in u[1][1]) if u[0] == 'run' else node_size[u[1]] for u in units]
        total = sum(sizes) or 1
        n_buckets = max(1, min(int(n_buckets), len(units)))
        self.n_buckets = n_buckets
        self.bucket_static = [[] for _ in range(n_buckets)]      # static (offset, n) segments
        self.bucket_static_params = [[] for _ in range(n_buckets)]
        self.bucket_nodes = [[] for _ in range(n_buckets)]
        self.bucket_of_param = {}
        acc = 0
        for u, sz in zip(units, sizes):
            b = min(n_buckets - 1, acc * n_buckets // total)
            if u[0] == 'run' and not u[1][0]:
                b = n_buckets - 1                                   # the stem always closes the exchange
            acc += sz
            if u[0] == 'run':
                idxs = u[1][1]
                self.bucket_static[b].append(span([fg.params[i] for i in idxs]))
                self.bucket_static_params[b] += idxs
                for i in idxs:
                    self.bucket_of_param[i] = b
            else:
                k = u[1]
                self.bucket_nodes[b].append(k)
                for ps in self.per_op[k]:
                    for p in ps:
                        self.bucket_of_param[fg.index[id(p)]] = b
        self.cand_span = [[span(ps) for ps in node] for node in self.per_op]
        self.cand_idx = [[[fg.index[id(p)] for p in ps] for ps in node] for node in self.per_op]
        self.cap = [sum(n_ for _, n_ in self.bucket_static[b]) + sum(node_size[k] for k in self.bucket_nodes[b])
                    for b in range(n_buckets)]
        self.staging = None
        self._segs = [None] * n_buckets
        self._pending = [0] * n_buckets
        self._launched = [True] * n_buckets
        self._armed = set()
        self._works = []
        self.comm_stream = torch.cuda.Stream() if (self.is_cuda and self.comm) else None
        self.inline = self.comm and _inline_collectives(group, self.is_cuda)
        self._tables = [None] * n_buckets
        if self.comm:
            for i, p in enumerate(fg.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))   # (see GradReducer: the only arrival signal)
        if self.is_cuda:
            self.fg.enable_sinks(None, self._arrived if self.comm else None, self if self.comm else None)   # (see GradReducer)
        self._mark_ev = {}

    # -- weight step --------------------------------------------------------------------------------------------
    def begin_weight_step(self):
        """After reset_binary_gates(): zero the buffer, give gradient views to the stem/head and the
        sampled candidates only (unsampled candidates keep grad=None, mixed.py:160-163), arm the buckets."""
        mops = self.net.redundant_modules
        self.fg.zero()
        if self.attach_all:
            p0 = self.fg.params[-1]
            if p0.grad is not self.fg.views[-1] or self.fg.params[0].grad is not self.fg.views[0]:
                self.fg.attach()
            else:
                self.fg.dirty = [True] * len(self.fg.params)
            self._active = None
        else:
            active = list(self.shared)
            for m, node in zip(mops, self.per_op):
                for i in m.active_index:
                    active += node[i]
            self.fg.attach(active)
            self._active = active
        if not self.comm:
            return
        armed = set()
        for b in range(self.n_buckets):
            segs = list(self.bucket_static[b])
            idxs = list(self.bucket_static_params[b])
            for k in self.bucket_nodes[b]:
                a = mops[k].active_index[0]
                if self.cand_span[k][a] is not None:
                    segs.append(self.cand_span[k][a])
                    idxs += self.cand_idx[k][a]
            self._segs[b] = segs
            self._tables[b] = self._segment_table(segs)   # (built here, while the host runs ahead of the GPU: the launch
            self._pending[b] = len(idxs)                   #  and the scatter at the end of the step are one call each)
            armed.update(idxs)
        self._armed = armed
        self._mark_ev = {}
        self._launched = [False] * self.n_buckets
        self._works = []
        if self.row_exchange is not None:
            self.row_exchange.begin()
        if self.staging is None:
            # ONE staging tensor, a slice per bucket: the scatter back into the flat buffer at the end of the step is then one
            # launch over all buckets' segments (round 5: one per bucket, all three behind the last all-reduce)
            caps = [_align(max(c, 64)) for c in self.cap]
            self.staging_all = torch.empty(sum(caps), dtype=torch.float32, device=self.fg.flat.device)
            self.staging_base = [sum(caps[:b]) for b in range(self.n_buckets)]
            self.staging = [self.staging_all[self.staging_base[b]:self.staging_base[b] + caps[b]] for b in range(self.n_buckets)]
        self._scatter_table = self._segment_table([seg for b in range(self.n_buckets) for seg in self._segs[b]],
                                                  [self.staging_base[b] + o for b in range(self.n_buckets) for o in self._seg_offsets(self._segs[b])])

    def _pending_params(self, b):
        return [i for i in self._armed if self.bucket_of_param.get(i) == b]

    def chain_marks(self, op_params):
        return _chain_marks(self, op_params, self.bucket_of_param, range(self.n_buckets))

    def exchanged_segments(self):
        """(offset, n) runs of the flat buffer that travel in this step's exchange (all buckets)."""
        return [seg for segs in self._segs if segs for seg in segs]

    def _arrived(self, i):
        if i not in self._armed:
            return
        self._armed.discard(i)
        self.fg.adopt(i)
        b = self.bucket_of_param[i]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _make_hook(self, i):
        def hook(_p):
            self._arrived(i)
        return hook

    def _launch(self, b, tail=False):
        if self._launched[b]:
            return
        self._launched[b] = True
        segs = self._segs[b]
        total = sum(n for _, n in segs)
        if total == 0:
            return
        stg = self.staging[b][:total]
        avg = _has_avg(self.group)
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        if self.is_cuda and tail and self.inline and _TAIL_ON_MAIN:
            # A bucket that is still open when backward has ended (always the stem's): nothing is left to overlap it with, so it
            # is exchanged on the CALLER's stream -- main -> collective -> main, two stream hand-overs instead of the four of
            # main -> communication stream -> collective -> communication stream -> main (each ~20 us of idle GPU at the very
            # end of the step: profiles/r05_timeline_search_vqa_dp1.txt)
            from . import ops
            ops.side_stream_barrier(torch.cuda.current_stream())
            self._pack(segs, stg, 0, table=self._tables[b])
            dist.all_reduce(stg, op=op, group=self.group)
            self._works.append(('inline', b, stg, not avg))
            return
        if self.is_cuda:
            from . import ops
            ev = self._mark_ev.pop(b, None)      # recorded inside the backbone chain, behind the bucket's last operator
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
            ops.side_stream_barrier(self.comm_stream)   # weight gradients the backbone chain put on its side stream
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._pack(segs, stg, 0, table=self._tables[b])
                if self.inline:       # pack -> all-reduce (-> scatter, finish_weight_step) in the order of this one stream
                    dist.all_reduce(stg, op=op, group=self.group)
                    w = 'inline'
                else:
                    w = dist.all_reduce(stg, op=op, group=self.group, async_op=True)
                if avg and not self.inline and os.environ.get('MMNAS_DP_EARLY_SCATTER', '0') == '1':
                    # Optional (MMNAS_DP_EARLY_SCATTER=1): scatter the averaged gradients back as soon as THIS bucket's
                    # all-reduce ends (RCCL: wait() only makes the communication stream wait), leaving only the last
                    # bucket's scatter behind the end of backward.  Measured in a one-rank group: +0.06 ms per step --
                    # the scatter kernels then run beside the backward's single-round GEMM launches and cost them more
                    # than the ~20 us they take off the tail.  Off by default.
                    w.wait()
                    self._pack(segs, stg, 1)
                    w = None
        else:
            self._pack(segs, stg, 0)
            w = dist.all_reduce(stg, op=op, group=self.group, async_op=True)
        self._works.append((w, b, stg, not avg))

    def finish_weight_step(self):
        """After backward: flush the buckets that are still open (always the stem's), wait, scatter the averaged
        gradients back into the flat buffer."""
        if not self.comm:
            self.fg.adopt_strays(None if self._active is None else [self.fg.index[id(p)] for p in self._active])
            return
        for i in list(self._armed):          # gradients that never arrived through a hook (e.g. produced under no hook)
            self.fg.adopt(i)
        self._armed = set()
        if self.row_exchange is not None:
            self.row_exchange.finish()
        for b in range(self.n_buckets):
            self._launch(b, tail=True)
        if self.is_cuda:
            one = _TAIL_ON_MAIN and all(w == 'inline' for w, _, _, _ in self._works) and len(self._works) == sum(1 for sg in self._segs if sg and sum(n for _, n in sg))
            if one:
                # every bucket's all-reduce is stream-ordered (inline): join the communication stream once, then ONE scatter
                # launch over all buckets' segments on the caller's stream
                torch.cuda.current_stream().wait_stream(self.comm_stream)
                need_div = self._works[0][3] if self._works else False
                nseg = sum(len(sg) for sg in self._segs if sg)
                if nseg:
                    from . import _lib as L
                    L.check(L.lib().mmnas_pack_segments_host(self._scatter_table, nseg, L.fptr(self.staging_all),
                                                             float(1.0 / self.world if need_div else 1.0), 1, L.stream()))
            else:
                with torch.cuda.stream(self.comm_stream):
                    for w, b, stg, need_div in self._works:
                        if w is None:
                            continue             # (scattered back right behind its all-reduce: _launch)
                        if w != 'inline':
                            w.wait()
                        self._pack(self._segs[b], stg, 1, 1.0 / self.world if need_div else 1.0, table=self._tables[b])
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w, b, stg, need_div in self._works:
                w.wait()
                self._pack(self._segs[b], stg, 1, 1.0 / self.world if need_div else 1.0)
        self._works = []

    @staticmethod
    def _seg_offsets(segs):
        off, out = 0, []
        for _, n in segs:
            out.append(off)
            off += n
        return out

    def _segment_table(self, segs, stg_offsets=None):
        """Kernel-argument table of (flat-buffer pointer, staging offset, length) per segment; stg_offsets: positions in the
        staging tensor (default: packed back to back from 0)."""
        if not self.is_cuda:
            return None
        from . import _lib as L
        arr = (L.Segment * max(len(segs), 1))()
        base = self.fg.flat.data_ptr()
        if stg_offsets is None:
            stg_offsets = self._seg_offsets(segs)
        for k, (o, n) in enumerate(segs):
            arr[k].ptr, arr[k].offset, arr[k].n = base + 4 * o, stg_offsets[k], n
        return arr

    def _pack(self, segs, stg, direction, scale=1.0, table=None):
        if self.is_cuda:
            from . import _lib as L
            arr = table if table is not None else self._segment_table(segs)
            # the table rides in the kernel arguments: no host->device copy, no stream synchronisation per step
            L.check(L.lib().mmnas_pack_segments_host(arr, len(segs), L.fptr(stg), float(scale), direction, L.stream()))
        else:  # CPU tensors (gloo tests): host plumbing only
            off = 0
            for o, n in segs:
                if direction == 0:
                    stg[off:off + n].copy_(self.fg.flat[o:o + n])
                else:
                    self.fg.flat[o:o + n].copy_(stg[off:off + n] * scale if scale != 1.0 else stg[off:off + n])
                off += n

    # -- arch step ----------------------------------------------------------------------------------------------
    def reduce_alpha_gate_grads(self):
        """Arch step: average dL/dgate over ranks (the only gradient the arch step uses, mixed.py:172).  When the
        gate gradients live in the net's flat [n_nodes, width] block (Net_Search.begin_arch_step) that block is
        all-reduced in place: one collective, no copies."""
        if not self.comm:
            return
        mops = self.net.redundant_modules
        block = None
        fl = getattr(self.net, '_flat_grads', None)
        if fl is not None and all(m.alpha_gate.grad is getattr(m.alpha_gate, '_mmnas_gate_grad', None) and
                                  m.alpha_gate.grad is not None for m in mops):
            block = fl[0]
        if block is None:   # gradients autograd allocated one by one (reference-style loop): stage them
            width = max(m.n_choices for m in mops)
            block = torch.zeros(len(mops), width, device=mops[0].alpha_gate.device)
            for i, m in enumerate(mops):
                if m.alpha_gate.grad is not None:
                    block[i, :m.n_choices] = m.alpha_gate.grad
        if _has_avg(self.group):
            dist.all_reduce(block, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(block, op=dist.ReduceOp.SUM, group=self.group)
            block.mul_(1.0 / self.world)
        if fl is None or block is not fl[0]:
            for i, m in enumerate(mops):
                if m.alpha_gate.grad is not None:
                    m.alpha_gate.grad.copy_(block[i, :m.n_choices])


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
