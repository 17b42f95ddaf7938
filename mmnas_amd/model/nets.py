"""Cells, backbones and the task networks (mmnas/model/{hygr,full}_{vqa,vgd,itm}.py) over the HIP
candidate operators.  One implementation serves the three tasks; the per-task modules
(hygr_vqa.py ... full_itm.py) export it under the reference's class names.  Module/attribute names
-- including the reference's ``backnone`` spelling (hygr_vqa.py:77) -- are kept so state_dicts are
interchangeable.

Stem and head: embedding + LSTM stay on torch (MIOpen); every Linear (imgfeat_linear, the relation
embeddings, AttFlat, the projections) and every LayerNorm run on the HIP GEMM / LayerNorm kernels.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, zeroterm
from ..utils.ops_adapter import OpsAdapter
from .mixed import MixedOp, sample_indices, sample_rows
from .modules import AttFlat, LayerNorm, RelHandle

OPS_ADAPTER = OpsAdapter()
LAZY_REL = os.environ.get('MMNAS_LAZY_REL', '1') != '0'   # 0: materialise rel embeddings as the reference does

# MCAN-style prior the supernet's alphas start from (hygr_vqa.py:138-156)
_PRIOR = {'enc': ['self_att_64', 'feed_forward'] * 6,
          'dec': ['rel_self_att_64', 'guided_att_64', 'feed_forward'] * 6}


class _Cell(nn.Module):
    def forward(self, s, pre=None, s_mask=None, pre_mask=None, rel_embed=None):
        # the reference writes sum(op(...) for op in node) (hygr_vqa.py:25): 0 + t is exact, so the
        # single-operator node skips that extra element-wise kernel
        for node in self.dag:
            outs = [op(s, pre, s_mask, pre_mask, rel_embed) for op in node]
            s = outs[0]
            for o in outs[1:]:
                s = s + o
        return s


class Cell_Search(_Cell):
    """NODES[type] sequential nodes, one MixedOp each (hygr_vqa.py:12-27)."""

    def __init__(self, __C, type):
        super().__init__()
        self.dag = nn.ModuleList(
            [nn.ModuleList([MixedOp(__C, type + '_safe')]) for _ in range(__C.NODES[type])])


class Cell_Full(_Cell):
    """Nodes built from __C.GENOTYPE[type]; a node may sum several operators (full_vqa.py:9-28)."""

    def __init__(self, __C, type):
        super().__init__()
        self.NODES = len(__C.GENOTYPE[type])
        self.dag = nn.ModuleList(
            [nn.ModuleList([OPS_ADAPTER.OPS[n](__C, norm=__C.OPS_NORM, residual=__C.OPS_RESIDUAL) for n in node])
             for node in __C.GENOTYPE[type]])


def _op_params(op):
    """An operator's parameter list (cached on the module: walking the module tree 30 times per step is host time)."""
    ps = op.__dict__.get('_mmnas_params')
    if ps is None:
        ps = op.__dict__['_mmnas_params'] = list(op.parameters())
    return ps


class _Backbone(nn.Module):
    CELL = None

    def __init__(self, __C):
        super().__init__()
        self.cells_enc = nn.ModuleList([self.CELL(__C, type='enc') for _ in range(__C.LAYERS)])
        self.cells_dec = nn.ModuleList([self.CELL(__C, type='dec') for _ in range(__C.LAYERS)])

    def _chain(self, x, y, x_mask, y_mask, x_rel_embed, y_rel_embed, packed=False):
        """The whole backbone as ONE native call per direction (ops.BackboneFn) -- possible when every node holds a
        single attention- or MLP-family operator, relation operators can take the lazy handle, and every parameter's
        gradient lives in a flat buffer the kernels may add into (a reducer / FlatAdam is attached).  Returns None
        when any of that does not hold: the per-operator path below serves every other case."""
        from .modules import FeedForward, FeedForward_deep, GuidedAtt, RelSelfAtt, SelfAtt
        if not (ops.chain_enabled() and x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32):
            return None
        mixed_mode = MixedOp.MODE in ('full', 'two')
        if MixedOp.MODE is not None and not (mixed_mode and ops.mixed_chain_enabled()):
            return None
        records, params, op_params = [], [], []
        mops = []
        # Two ways for the parameter gradients to leave the chain: into a flat gradient buffer the parameters' .grad are views
        # of (a reducer / FlatAdam owns them: "sinks"), or -- plain autograd use: the unchanged scripts under stock DDP -- as
        # autograd outputs of the node (ops.autograd_chain_enabled: the kernels accumulate into a per-call buffer whose views
        # backward returns).  Decided by the first operator's first parameter; the arch step's mixed chain needs the sinks.
        mode_ag = [None]      # decided by the FIRST operator the chain takes (an unsampled candidate's parameters tell nothing)

        def autograd_mode(first_param):
            if mode_ag[0] is None:
                mode_ag[0] = not ops._sinked((first_param,))
            return mode_ag[0]

        def record(op, on_y, rel):
            """(ChainOp, parameters) of one operator, or None when the chain cannot take it.  Autograd mode: the ChainOp is a
            cached template without gradient pointers (patched below, once the per-call buffer exists)."""
            t = type(op)
            if t is SelfAtt or t is GuidedAtt or t is RelSelfAtt:
                if t is GuidedAtt and not on_y:
                    return None
                rh = None
                if t is RelSelfAtt:
                    if not (isinstance(rel, RelHandle) and rel.fusable(op.mhatt.linear_r.weight.shape[0])):
                        return None
                    rh = rel
                if autograd_mode(op.mhatt.linear_q.weight):
                    if mixed_mode or not ops.autograd_chain_enabled() or not torch.is_grad_enabled():
                        return None
                    return ops.chain_att_template(op, on_y, t is not GuidedAtt, rh)
                ps = _op_params(op)
                if not ops._sinked(ps) or (rh is not None and not ops._sinked((rh.weight, rh.bias))):
                    return None
                return ops.chain_att_record_cached(op, on_y, t is not GuidedAtt, rh)
            if t is FeedForward or t is FeedForward_deep:
                m = op.mlp
                if t is FeedForward:
                    ws, bs = [m.fc.linear.weight, m.linear.weight], [m.fc.linear.bias, m.linear.bias]
                else:
                    ws = [op.fc.linear.weight, m.fc.linear.weight, m.linear.weight]
                    bs = [op.fc.linear.bias, m.fc.linear.bias, m.linear.bias]
                if autograd_mode(ws[0]):
                    if mixed_mode or not ops.autograd_chain_enabled() or not torch.is_grad_enabled():
                        return None
                    return ops.chain_mlp_template(op, on_y, ws, bs)
                ps = _op_params(op)
                if not ops._sinked(ps):
                    return None
                return ops.chain_mlp_record_cached(op, on_y, ws, bs)
            return None

        for on_y, cells in ((0, self.cells_enc), (1, self.cells_dec)):
            rel = y_rel_embed if on_y else x_rel_embed
            for cell in cells:
                for node in cell.dag:
                    if len(node) != 1:
                        return None
                    op = node[0]
                    if mixed_mode:
                        # architecture step: every evaluated candidate of the node, in candidate order; only the sampled
                        # one is differentiated (mixed.py:59-68)
                        if not isinstance(op, MixedOp) or len(op.active_index) != 1:
                            return None
                        k = len(mops)
                        mops.append(op)
                        for i in sorted(op.active_index + op.inactive_index):
                            cand = op.candidate_ops[i]
                            if cand is None:
                                return None
                            got = record(cand, on_y, rel)
                            if got is None:
                                return None
                            rec, used = got
                            rec.node, rec.cand, rec.detached = k, i, int(i != op.active_index[0])
                            records.append(rec)
                            if not rec.detached:
                                params += used
                        continue
                    if isinstance(op, MixedOp):
                        op = op.active_op
                    got = record(op, on_y, rel)
                    if got is None:
                        return None
                    rec, used = got
                    rec.node, rec.cand, rec.detached = len(records), 0, 0
                    records.append(rec)
                    params += used
                    op_params.append(used)
        gviews, ptensors = None, ()
        if mode_ag[0]:
            # the per-call gradient buffer: one zero fill, one view per distinct parameter (the shared relation stem is listed
            # by every relation operator), the descriptors patched to point there; seeds drawn in operator order as always
            if not records or any(not p.requires_grad for p in params):
                return None
            uniq = list({id(p): p for p in params}.values())
            sizes = [(p.numel() + 63) // 64 * 64 for p in uniq]
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=x.device)
            views, off = {}, 0
            for p, n in zip(uniq, sizes):
                views[id(p)] = flat[off:off + p.numel()].view(p.shape)
                off += n
            gp = lambda p: views[id(p)].data_ptr()
            patched = []
            for tmpl, used in zip(records, op_params):
                rec = ops.patched_record(tmpl, used, gp)
                rec.node, rec.cand, rec.detached = len(patched), 0, 0
                patched.append(rec)
            records = patched
            gviews, ptensors = [views[id(p)] for p in uniq], tuple(uniq)
            op_params = None          # (bucket marks are a reducer's: none here)
        mixed = None
        if mixed_mode:
            # the nodes' binary gates / gate gradients must be rows of two [n_nodes, width] blocks (Net_Search._flat_alphas,
            # begin_arch_step): the chain addresses them by node index
            g0 = mops[0].alpha_gate
            gr0 = g0.grad
            if gr0 is None or gr0 is not getattr(g0, '_mmnas_gate_grad', None):
                return None
            width = max(m.n_choices for m in mops)
            for k, m in enumerate(mops):
                g = m.alpha_gate
                if (g.grad is None or g.grad is not getattr(g, '_mmnas_gate_grad', None) or
                        g.data_ptr() != g0.data_ptr() + 4 * width * k or g.grad.data_ptr() != gr0.data_ptr() + 4 * width * k):
                    return None
            mixed = (g0.data_ptr(), gr0.data_ptr(), width)
            op_params = None
        if op_params is not None:
            # The relation-bias backward of every lazy-handle relation operator of a stream is ONE launch behind the stream's
            # first operator (its last in backward order; relmulti.hip, ops.hip chain_rel_bwd): that operator completes the
            # relation parameters' gradients -- linear_r of each relation operator and the shared stem layer -- so a
            # reducer's bucket mark (dp._chain_marks: "the operator that issues a parameter's last gradient") must see them
            # there, not at the relation operator itself.  (Holds with MMNAS_REL_HOIST=0 too: a later mark is never wrong.)
            first = {}
            for i, rec in enumerate(records):
                first.setdefault(rec.on_y, i)
            moved = {}
            for i, rec in enumerate(records):
                if rec.kind == 0 and rec.att.flags & ops.L.F_REL and i != first[rec.on_y]:
                    moved.setdefault(first[rec.on_y], []).extend(op_params[i][-4:])
                    op_params[i] = op_params[i][:-4]
            for i, ps in moved.items():
                op_params[i] = list(op_params[i]) + ps
            # Likewise the key / value projections of the guided operators: their weight gradients (and the key / value source
            # gradient) are issued by grouped launches behind the FIRST guided operator of the chain, the last in backward
            # order (ops.hip chain_guided_kv_bwd).  chain_att_record: params = [Wq, Wk, Wv, Wm, ...].
            guided = [i for i, rec in enumerate(records) if rec.kind == 0 and not (rec.att.flags & ops.L.F_SELF)]
            if len(guided) > 1:
                g0 = guided[0]
                extra = []
                for i in guided[1:]:
                    extra += list(op_params[i][1:3])
                    op_params[i] = [op_params[i][0]] + list(op_params[i][3:])
                op_params[g0] = list(op_params[g0]) + extra
        if not records or len(records) > 128:
            return None
        xr = x_rel_embed.raw if isinstance(x_rel_embed, RelHandle) else None
        yr = y_rel_embed.raw if isinstance(y_rel_embed, RelHandle) else None
        # ragged decoder stream (ops.Ragged; MMNAS_UNPAD / ops.set_unpad): the net's forward attached the description of the
        # batch to the mask it derived from the region features.  Needs the lazy relation handle (raw boxes) when a relation
        # operator is sampled.
        ragged = getattr(y_mask, '_mmnas_ragged', None) if y_mask is not None else None
        if ragged is not None and y_rel_embed is not None and not isinstance(y_rel_embed, RelHandle):
            ragged = None
        if packed and ragged is None:
            return None           # (packed image rows only make sense on the ragged stream: the caller falls back to padded rows)
        return ops.backbone_chain(x, y, x_mask, y_mask, xr, yr, records, params, op_params, mixed, ragged, packed, gviews, ptensors)

    def chain_packed(self, x, y_packed, x_mask, y_mask, x_rel_embed, y_rel_embed):
        """The ragged decoder stream end to end: `y_packed` [sum n_b, d] are the projected VALID region rows (the stem ran
        on them only), the decoder output comes back packed (for the head's AttFlat over packed rows).  None when the chain
        or the ragged stream cannot take this call -- the caller then projects the padded rows and calls forward()."""
        return self._chain(x, y_packed, x_mask, y_mask, x_rel_embed, y_rel_embed, packed=True)

    def forward(self, x, y, x_mask, y_mask, x_rel_embed, y_rel_embed):
        out = self._chain(x, y, x_mask, y_mask, x_rel_embed, y_rel_embed)
        if out is not None:
            return out
        # encoder cells over the language stream, then decoder cells over the image stream with
        # pre = final language state (hygr_vqa.py:45-52)
        for cell in self.cells_enc:
            x = cell(s=x, s_mask=x_mask, rel_embed=x_rel_embed)
        for cell in self.cells_dec:
            y = cell(s=y, pre=x, s_mask=y_mask, pre_mask=x_mask, rel_embed=y_rel_embed)
        return x, y


class Backbone_Search(_Backbone):
    CELL = Cell_Search


class Backbone_Full(_Backbone):
    CELL = Cell_Full


def make_mask(feature):
    """True where a whole feature row is zero = padding (hygr_vqa.py:121-122).  Float features on the GPU (the
    [B,100,2048] region tensor) take one pass of mmnas_row_is_zero; token indices keep the torch expression."""
    if feature.dtype == torch.float32 and feature.is_cuda:
        return ops.row_is_zero(feature).unsqueeze(1).unsqueeze(2)
    if feature.shape[-1] == 1:   # token indices [B, S, 1]: sum(|x|) == 0 is x == 0 (one kernel instead of three)
        return (feature.squeeze(-1) == 0).unsqueeze(1).unsqueeze(2)
    return (torch.sum(torch.abs(feature), dim=-1) == 0).unsqueeze(1).unsqueeze(2)


class _Net(nn.Module):
    TASK = 'vqa'
    SEARCH = False

    def _build(self, __C, init_dict):
        self._cfg = __C
        d = __C.HSIZE
        self.embedding = nn.Embedding(num_embeddings=init_dict['token_size'], embedding_dim=__C.WORD_EMBED_SIZE)
        self.embedding.weight.data.copy_(torch.from_numpy(init_dict['pretrained_emb']))
        # its gradient is the rows of the batch's tokens: data-parallel reducers exchange those rows (dp.RowExchange)
        self.embedding.weight._mmnas_row_sparse = True
        self.lstm = nn.LSTM(input_size=__C.WORD_EMBED_SIZE, hidden_size=d, num_layers=1, batch_first=True)
        feat = __C.FRCNFEAT_SIZE
        if __C.BBOX_FEATURE:
            self.bboxfeat_linear = nn.Linear(5, __C.BBOXFEAT_EMB_SIZE)
            feat += __C.BBOXFEAT_EMB_SIZE
        self.imgfeat_linear = nn.Linear(feat, d)
        self.backnone = (Backbone_Search if self.SEARCH else Backbone_Full)(__C)
        self.attflat_x = AttFlat(__C)
        if self.TASK == 'vgd':
            self.attfc_y = nn.Linear(d, __C.ATTFLAT_OUT_SIZE)      # full_vgd.py:78
        else:
            self.attflat_y = AttFlat(__C)
        self.proj_norm = LayerNorm(__C.ATTFLAT_OUT_SIZE)
        if self.TASK == 'vgd':
            self.proj_scores = nn.Linear(__C.ATTFLAT_OUT_SIZE, 1)
            self.proj_reg = nn.Linear(__C.ATTFLAT_OUT_SIZE, 4)
        elif self.TASK == 'itm':
            self.proj = nn.Linear(__C.ATTFLAT_OUT_SIZE, 1)
        else:
            self.proj = nn.Linear(__C.ATTFLAT_OUT_SIZE, init_dict['ans_size'])
        # only the VQA / ITM supernets embed the token relations (hygr_vqa.py:83, hygr_itm.py:77;
        # hygr_vgd.py has linear_y_rel only; the fixed-arch nets never do)
        if self.SEARCH and self.TASK in ('vqa', 'itm'):
            self.linear_x_rel = nn.Linear(3, __C.REL_SIZE)
        self.linear_y_rel = nn.Linear(4, __C.REL_SIZE)

    @staticmethod
    def make_mask(feature):
        return make_mask(feature)

    def forward(self, input):
        frcn_feat, bbox_feat, y_rel_embed, ques_ix, x_rel_embed = input
        C = self._cfg
        x_mask = make_mask(ques_ix.unsqueeze(2))
        y_mask = make_mask(frcn_feat)
        if self.TASK != 'vgd' and ops.unpad_enabled():
            # ragged decoder stream: only where the head drops the padding rows (AttFlat's mask; the grounding head scores
            # EVERY region row, padding included -- full_vgd.py:105-114 -- so it keeps the padded computation)
            y_mask._mmnas_ragged = ops.ragged_info_for(frcn_feat, y_mask)
        emb = ops.embedding(ques_ix, self.embedding)
        # (one persistent launch per pass: ops.LstmFn; nn.LSTM = MIOpen only for shapes outside its range)
        x_in = ops.lstm(emb, self.lstm) if (ops.lstm_enabled() and ops.lstm_supported(emb, self.lstm)) else self.lstm(emb)[0]
        if C.BBOX_FEATURE:
            bb = ops.linear(bbox_feat, self.bboxfeat_linear.weight, self.bboxfeat_linear.bias)
            frcn_feat = torch.cat((frcn_feat, bb), dim=-1)
        # relation embeddings relu(linear_*_rel(raw)) (hygr_vqa.py:110-111): passed down as lazy handles
        # -- the [B,S,S,64] tensors are only materialised if an operator asks for a plain tensor
        if LAZY_REL:
            if hasattr(self, 'linear_x_rel'):
                x_rel_embed = RelHandle(x_rel_embed, self.linear_x_rel.weight, self.linear_x_rel.bias)
            y_rel_embed = RelHandle(y_rel_embed, self.linear_y_rel.weight, self.linear_y_rel.bias)
        else:
            if hasattr(self, 'linear_x_rel'):
                x_rel_embed = ops.linear(x_rel_embed, self.linear_x_rel.weight, self.linear_x_rel.bias, relu=True)
            y_rel_embed = ops.linear(y_rel_embed, self.linear_y_rel.weight, self.linear_y_rel.bias, relu=True)
        # Ragged decoder stream end to end (round 5): the stem projects the VALID region rows only (packed), the backbone
        # chain runs on them, the head's AttFlat pools over each sample's own rows -- no padded row is computed anywhere
        # (hygr_vqa.py:97-122 computes them and masks them out).  Engages when the native head and the chain both do.
        rg = getattr(y_mask, '_mmnas_ragged', None)
        packed = None
        native_head = self.TASK != 'vgd' and ops.chain_enabled() and frcn_feat.is_cuda
        # (head_record() draws the head's dropout seeds: it stays BEHIND the backbone call, where the per-operator path draws
        #  them too -- the seed order is what lets dropout be replayed across the paths)
        if rg is not None and native_head and ops._sinked((self.attflat_x.mlp.fc.linear.weight,)):
            y_pk = ops.linear(ops.pack_rows_fn(frcn_feat, rg), self.imgfeat_linear.weight, self.imgfeat_linear.bias)
            packed = self.backnone.chain_packed(x_in, y_pk, x_mask, y_mask, x_rel_embed, y_rel_embed)
        if packed is not None:
            x_out, y_out = packed
            hd, hp = ops.head_record(self.attflat_x, self.attflat_y, self.proj_norm, self.proj, self.training)
            if hd is not None:
                out = ops.HeadFn.apply(x_out, y_out, x_mask, y_mask, hd, hp, rg)
                return torch.sigmoid(out.squeeze(-1)) if self.TASK == 'itm' else out
            y_out = ops.unpack_rows_fn(y_out, rg, frcn_feat.shape[0], frcn_feat.shape[1])   # (no native head after all)
        else:
            y_in = ops.linear(frcn_feat, self.imgfeat_linear.weight, self.imgfeat_linear.bias)
            x_out, y_out = self.backnone(x_in, y_in, x_mask, y_mask, x_rel_embed, y_rel_embed)
            if native_head and x_out.is_cuda:
                # AttFlat x 2 + proj_norm + proj as one native call per direction (needs the flat-gradient sinks)
                hd, hp = ops.head_record(self.attflat_x, self.attflat_y, self.proj_norm, self.proj, self.training)
                if hd is not None:
                    out = ops.HeadFn.apply(x_out, y_out, x_mask, y_mask, hd, hp)
                    return torch.sigmoid(out.squeeze(-1)) if self.TASK == 'itm' else out
        x_out = self.attflat_x(x_out, x_mask)
        if self.TASK == 'vgd':  # per-object scores + box regression (full_vgd.py:105-114)
            y_out = ops.linear(y_out, self.attfc_y.weight, self.attfc_y.bias)
            xy = self.proj_norm(x_out.unsqueeze(1) + y_out)
            scores = ops.linear(xy, self.proj_scores.weight, self.proj_scores.bias).squeeze(-1)
            if C.SCORES_LOSS == 'kld':
                scores = F.log_softmax(scores, dim=-1)
            return scores, ops.linear(xy, self.proj_reg.weight, self.proj_reg.bias)
        y_out = self.attflat_y(y_out, y_mask)
        xy = self.proj_norm(x_out + y_out)
        out = ops.linear(xy, self.proj.weight, self.proj.bias)
        if self.TASK == 'itm':  # matching score (full_itm.py:109-112)
            return torch.sigmoid(out.squeeze(-1))
        return out


class NetFullBase(_Net):
    """Fixed architecture read from __C.GENOTYPE (full_vqa.py:56-114)."""
    SEARCH = False

    def __init__(self, __C, init_dict):
        super().__init__()
        self._build(__C, init_dict)
        zeroterm.adopt(self)      # the scripts' `0 * sum(p.sum() ...)` line (train_vqa.py:299) as ONE autograd node


class NetSearchBase(_Net):
    """Supernet (hygr_vqa.py:55-297): every cell node is a MixedOp."""
    SEARCH = True

    def __init__(self, __C, init_dict):
        super().__init__()
        self._redundant_modules = None
        self._unused_modules = None
        self._build(__C, init_dict)
        self.init_arch()
        self._net_weights = [(n, p) for n, p in self.named_parameters()
                             if 'alpha_prob' not in n and 'alpha_gate' not in n]
        self._flat = None
        self._flat_grads = None
        self._probs_cache = None
        zeroterm.adopt(self)      # the scripts' three `0 * sum(p.sum() ...)` lines (search_vqa.py:285-288) as three autograd nodes

    # -- architecture parameters ------------------------------------------------------------
    def init_arch(self):
        C = self._cfg
        self._alphas_prob = [(n, p) for n, p in self.named_parameters() if 'alpha_prob' in n]
        self._alphas_gate = [(n, p) for n, p in self.named_parameters() if 'alpha_gate' in n]
        for p in self.alpha_prob_parameters():
            if C.ALPHA_INIT_TYPE == 'normal':
                p.data.normal_(0, 1e-3)
            elif C.ALPHA_INIT_TYPE == 'uniform':
                p.data.uniform_(-1e-3, 1e-3)
        # ... then overwritten by the +1/-1 prior, as the reference does (hygr_vqa.py:142-156)
        prior = _PRIOR['enc'][:12] + _PRIOR['dec']
        for ix, (name, (_, p)) in enumerate(zip(prior, self._alphas_prob)):
            space = OPS_ADAPTER.Used_OPS['enc_safe' if ix < 12 else 'dec_safe']
            v = np.full(len(space), -1.0, dtype=np.float32)
            v[space.index(name)] = 1.0
            p.data = torch.from_numpy(v)

    @property
    def redundant_modules(self):
        if self._redundant_modules is None:
            self._redundant_modules = [m for m in self.modules() if isinstance(m, MixedOp)]
        return self._redundant_modules

    def _flat_alphas(self):
        """Re-home every node's alpha_prob / alpha_gate as views of two [n_nodes, 4] device buffers so
        sampling, gate writes and the alpha-gradient are one batched operation per step."""
        mops = self.redundant_modules
        dev = mops[0].alpha_prob.device
        if self._flat is not None and self._flat[0].device == dev and \
                mops[0].alpha_prob.data_ptr() == self._flat[0].data_ptr():
            return self._flat
        n = len(mops)
        width = max(m.n_choices for m in mops)
        prob = torch.full((n, width), float('-inf'), device=dev)
        gate = torch.zeros((n, width), device=dev)
        for i, m in enumerate(mops):
            prob[i, :m.n_choices] = m.alpha_prob.data
            gate[i, :m.n_choices] = m.alpha_gate.data
            m.alpha_prob.data = prob[i, :m.n_choices]
            m.alpha_gate.data = gate[i, :m.n_choices]
        self._flat = (prob, gate)
        self._flat_grads = (torch.zeros_like(gate), torch.zeros_like(gate))   # dL/dgate, dL/dalpha blocks
        self._probs_cache = None
        return self._flat

    def begin_arch_step(self):
        """Arch step, before forward: zero the [n_nodes, width] gate-gradient block and make every node's
        alpha_gate.grad a row of it -- the fused gated-sum backward (ops.MixedSumFn) then adds the gate gradients
        straight into the block, which is what the data-parallel exchange all-reduces and the fused architecture
        update (harness.ArchAdam) reads.  Returns (gate_grad, prob_grad)."""
        self._flat_alphas()
        gg, pg = self._flat_grads
        gg.zero_()
        for i, m in enumerate(self.redundant_modules):
            row = gg[i, :m.n_choices]
            m.alpha_gate.grad = row
            m.alpha_gate._mmnas_gate_grad = row
        return gg, pg

    def _probs_cpu(self, prob):
        """softmax(alpha_prob) of every node on the host.  The alphas change only at architecture steps (one in six
        of the bilevel loop, search_vqa.py:149-150), so the device->host copy -- a full stream synchronisation -- is
        made only when they did: the cache keys on the parameters' version counters (in-place optimizer updates and
        load_state_dict bump them) plus MixedOp.alpha_version for the `.data` writes of the rescale step."""
        key = tuple((m.alpha_prob._version, m.alpha_version) for m in self.redundant_modules)
        if self._probs_cache is None or self._probs_cache[0] != key:
            self._probs_cache = (key, torch.softmax(prob.detach(), dim=1).cpu())
        return self._probs_cache[1]

    def invalidate_arch_cache(self):
        """Call after writing alpha_prob through `.data` (anything that does not bump the tensor version)."""
        self._probs_cache = None

    def reset_binary_gates(self):
        """binarize() every node (hygr_vqa.py:168-173): sampled on the host from the cached probabilities, the gates
        written by one kernel that carries the indices in its arguments -- no copy in either direction, so the host
        is free to run ahead of the GPU."""
        prob, gate = self._flat_alphas()
        probs = self._probs_cpu(prob)
        mops = self.redundant_modules
        idx = (ctypes.c_int * len(mops))()
        # a reducer that keeps every gradient view attached (dp.SupernetReducer(attach_all=True)) zeroes the buffer itself
        keep = getattr(self, 'keep_candidate_grads', False)
        if MixedOp.MODE is None:
            # one draw for all nodes (rows of the cached probability matrix; padding columns have probability 0):
            # 30 separate multinomial calls were 0.3 ms of host time per step
            drawn = sample_rows(probs)
            for i, m in enumerate(mops):
                a = drawn[i]
                m.set_active([a], [j for j in range(m.n_choices) if j != a], write_gate=False)
                idx[i] = a
                if not keep:
                    m.clear_candidate_grads()
        else:
            for i, m in enumerate(mops):
                act, inact = sample_indices(probs[i, :m.n_choices], MixedOp.MODE)
                m.set_active(act, inact, write_gate=False)
                idx[i] = act[0]
                if not keep:
                    m.clear_candidate_grads()
        if gate.is_cuda and len(mops) <= 128:
            from .. import _lib as L
            L.check(L.lib().mmnas_onehot_rows(L.fptr(gate), gate.shape[0], gate.shape[1], idx, L.stream()))
        else:
            g = torch.zeros(gate.shape)
            for i in range(len(mops)):
                g[i, idx[i]] = 1.0
            gate.copy_(g)

    def set_sampled(self, plan):
        """Install an explicit list of (active, inactive) choices, one per node (tests, replay)."""
        _, gate = self._flat_alphas()
        g = torch.zeros(gate.shape)
        for i, (m, (act, inact)) in enumerate(zip(self.redundant_modules, plan)):
            m.set_active(act, inact, write_gate=False)
            g[i, act[0]] = 1.0
        gate.copy_(g)

    def unused_modules_off(self):
        """Temporarily replace the candidates that take no part in this step by None (hygr_vqa.py:175-187)."""
        # (the entries are swapped in the ModuleList's own dict: `candidate_ops[i] = None` goes through nn.Module.__setattr__,
        #  ~3 us a time, 150 times per step in both directions -- 0.45 ms of the unchanged loop's host time)
        self._unused_modules = []
        full = MixedOp.MODE in ('full', 'two')
        for m in self.redundant_modules:
            involved = m.active_index + (m.inactive_index if full else [])
            mods = m.candidate_ops._modules
            unused = {}
            for i in range(m.n_choices):
                if i not in involved:
                    k = str(i)
                    unused[k] = mods[k]
                    mods[k] = None
            self._unused_modules.append(unused)

    def unused_modules_back(self):
        if self._unused_modules is None:
            return
        for m, unused in zip(self.redundant_modules, self._unused_modules):
            m.candidate_ops._modules.update(unused)
        self._unused_modules = None

    def set_arch_param_grad(self):
        for m in self.redundant_modules:
            m.set_arch_param_grad()

    def rescale_updated_arch_param(self):
        for m in self.redundant_modules:
            m.rescale_updated_arch_param()

    def set_chosen_op_active(self):
        for m in self.redundant_modules:
            m.set_chosen_op_active()

    # -- parameter views used by the search loop (hygr_vqa.py:218-240) -------------------------
    def alpha_prob_parameters(self):
        for _, p in self._alphas_prob:
            yield p

    def alpha_gate_parameters(self):
        for _, p in self._alphas_gate:
            yield p

    def named_alpha_prob_parameters(self):
        return iter(self._alphas_prob)

    def named_alpha_gate_parameters(self):
        return iter(self._alphas_gate)

    def net_parameters(self):
        for _, p in self._net_weights:
            yield p

    def named_net_parameters(self):
        return iter(self._net_weights)

    # -- read-out (hygr_vqa.py:242-297) ----------------------------------------------------------
    def genotype(self):
        gene = {'enc': [], 'dec': []}
        for n, p in self._alphas_prob:
            kind = 'enc' if 'cells_enc' in n else 'dec'
            gene[kind].append([OPS_ADAPTER.Used_OPS[kind][int(torch.argmax(p.data))]])
        return gene

    def genotype_weights(self):
        w = {'w_enc': [], 'w_dec': []}
        with torch.no_grad():
            for n, p in self._alphas_prob:
                w['w_enc' if 'cells_enc' in n else 'w_dec'].append(F.softmax(p, dim=-1).cpu().numpy())
        return w
