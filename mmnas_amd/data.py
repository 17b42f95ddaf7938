"""Data path feeding the operator hot path: what the reference's `mmnas/loader/load_data_*.py`
do per sample on the CPU and then ship through DDP's scatter every step (search_vqa.py:271,282), arranged so that
only the small tensors cross PCIe and the copies overlap the previous step.

  * `pad_rows` / `bbox_features`: the loaders' `proc_img_feat` (load_data_vqa.py:252-263) and `proc_bbox_feat`
    (load_data_vqa.py:266-275) restated for whole batches.
  * `load_frcn_npz`: one region-feature file (`x` [d, n] transposed, `bbox` [n, 4], `image_h`, `image_w`) ->
    padded features, 5-d box features, raw boxes and the box count (load_data_vqa.py:224-235).
  * `collate_regions`: a list of such samples -> batch arrays; the [B,100,100,4] relation tensor is NOT built on the
    host: `relations_on_device` computes it on the GPU from the [B,100,4] boxes (ops.relation_embedding), so a
    batch uploads 52 MB of features + 6 KB of boxes instead of 52 MB + 10 MB.
  * `DevicePrefetcher`: pinned staging buffers + a copy stream; batch i+1 is uploaded while step i computes.

  * `tokenize`: `proc_ques` (load_data_vqa.py:278-296); `semantic_relations_on_device`: `semantic_embedding`
    (load_data_vqa.py:36-58) for a whole batch on the GPU from the token indices and the GloVe table.

Parity PINNED: the reference's loader modules import `en_vectors_web_lg` / `spacy` at module level and cannot be
imported, so tests/golden/make_golden.py compiles the functions themselves out of the loader's source (`ast`) in the
build container and stores their outputs on deterministic inputs (tests/golden/loader.npz); tests/test_data.py checks
every function here against them.  The spaCy vocabulary itself and the answer scoring stay out of scope.
"""
import re

import numpy as np
import torch


def pad_rows(feat, pad_size):
    """proc_img_feat (load_data_vqa.py:252-263): keep at most pad_size rows, zero-pad to exactly pad_size."""
    feat = np.asarray(feat)
    if feat.shape[0] > pad_size:
        feat = feat[:pad_size]
    out = np.zeros((pad_size,) + feat.shape[1:], dtype=feat.dtype)
    out[:feat.shape[0]] = feat
    return out


def bbox_features(bbox, img_shape):
    """proc_bbox_feat (load_data_vqa.py:266-275): (x1/w, y1/h, x2/w, y2/h, box area / image area), float32 [n,5];
    img_shape = (height, width)."""
    bbox = np.asarray(bbox, dtype=np.float32)
    h, w = float(img_shape[0]), float(img_shape[1])
    out = np.zeros((bbox.shape[0], 5), dtype=np.float32)
    out[:, 0] = bbox[:, 0] / w
    out[:, 1] = bbox[:, 1] / h
    out[:, 2] = bbox[:, 2] / w
    out[:, 3] = bbox[:, 3] / h
    out[:, 4] = (bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1]) / (h * w)
    return out


def load_frcn_npz(path_or_file, pad_size=100):
    """One bottom-up-attention feature file -> dict(frcn_feat [pad,d], bbox_feat [pad,5], bbox [pad,4], nobj)."""
    z = np.load(path_or_file)
    x = z['x'].transpose((1, 0)).astype(np.float32)
    bbox = z['bbox'].astype(np.float32)
    n = min(bbox.shape[0], pad_size)
    return {'frcn_feat': pad_rows(x, pad_size),
            'bbox_feat': pad_rows(bbox_features(bbox, (z['image_h'], z['image_w'])), pad_size),
            'bbox': pad_rows(bbox, pad_size), 'nobj': np.int32(n)}


def collate_regions(samples):
    """list of load_frcn_npz() dicts -> dict of stacked arrays (frcn_feat [B,S,d], bbox_feat [B,S,5], bbox [B,S,4],
    nobj [B] int32)."""
    return {'frcn_feat': np.stack([s['frcn_feat'] for s in samples]),
            'bbox_feat': np.stack([s['bbox_feat'] for s in samples]),
            'bbox': np.stack([s['bbox'] for s in samples]),
            'nobj': np.asarray([s['nobj'] for s in samples], dtype=np.int32)}


def tokenize(question, token_to_ix, max_token=14):
    """proc_ques (load_data_vqa.py:278-296): lower-case, strip punctuation, '-' and '/' become spaces, unknown words map
    to token_to_ix['UNK'], zero padding.  Returns (ques_ix int64 [max_token], min(#words, max_token))."""
    words = re.sub(r"([.,'!?\"()*#:;])", '', question.lower()).replace('-', ' ').replace('/', ' ').split()
    ix = np.zeros(max_token, np.int64)
    for i, w in enumerate(words[:max_token]):
        ix[i] = token_to_ix.get(w, token_to_ix['UNK'])
    return ix, min(len(words), max_token)


_ANS_SCORE = (0.0, 0.3, 0.6, 0.9)


def answer_targets(answer_lists, ans_to_ix, normalize=None):
    """DataSet.proc_ans / get_score (load_data_vqa.py:299-333) for a batch: [B, len(ans_to_ix)] float32 soft targets
    from each question's annotator answers (count -> 0 / .3 / .6 / .9 / 1).  `normalize` is the answer normaliser the
    loader applies first (the reference's `preprocess_answer`, the VQA evaluation script's table-driven clean-up, is
    not part of this package: pass it in, or pass already-normalised strings)."""
    out = np.zeros((len(answer_lists), len(ans_to_ix)), np.float32)
    for b, answers in enumerate(answer_lists):
        counts = {}
        for a in answers:
            if normalize is not None:
                a = normalize(a)
            counts[a] = counts.get(a, 0) + 1
        for a, n in counts.items():
            j = ans_to_ix.get(a)
            if j is not None:
                out[b, j] = _ANS_SCORE[n] if n < 4 else 1.0
    return out


def semantic_relations_on_device(ques_ix, nwords, emb):
    """[B,S] token indices + [B] word counts + the [V,300] GloVe table (device tensors) -> the loaders' zero-padded
    [B,S,S,3] token-relation tensor (semantic_embedding, load_data_vqa.py:36-58), on the GPU."""
    from . import ops
    return ops.semantic_embedding(ques_ix, nwords, emb)


def relations_on_device(bbox, nobj):
    """[B,S,4] boxes + [B] counts (device tensors) -> the loaders' zero-padded [B,S,S,4] relation tensor, on the GPU."""
    from . import ops
    return ops.relation_embedding(bbox, nobj)


class DevicePrefetcher:
    """Wraps an iterable of batches (tuples / lists / dicts of CPU tensors or numpy arrays) and yields the same
    structure on `device`.  Each batch is staged in pinned memory and copied on a side stream while the previous
    batch is being consumed; the consumer's stream waits on the copy before the tensors are handed out, and the
    tensors are recorded on it so the caching allocator does not reuse them early.  On a CPU `device` it is a
    plain pass-through (tests).

    lengths=(features_key, lengths_key): the batch element `lengths_key` (a CPU int tensor / array with the number of
    detected regions per sample, which the loaders know: load_data_vqa.py:221-246 pads behind them) is attached to the
    DEVICE copy of element `features_key` as `_mmnas_lengths` -- what ops.ragged_info_for reads when the ragged decoder
    stream is on (ops.set_unpad), so a fresh batch per step needs no device-to-host copy to learn its lengths.
    CONTRACT (the loaders', load_data_vqa.py:221-246: `proc_img_feat` zero-pads behind the detected boxes): sample b's rows
    0 .. n_b - 1 are its regions (non-zero rows) and rows n_b .. S - 1 are all-zero -- the same rows the networks' masks
    (make_mask: all-zero feature rows) call padding.  The counts are checked against the HOST copy of the features before
    the upload, because the ragged stream trusts them: `validate` = 'boundary' (default: row n_b - 1 non-zero, row n_b
    all-zero -- two rows per sample), 'full' (every row: one pass over the batch on the host) or None."""

    def __init__(self, loader, device, lengths=None, validate='boundary'):
        self.loader = loader
        self.lengths = lengths
        self.validate = validate
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None

    def _to_tensor(self, a):
        return torch.from_numpy(a) if isinstance(a, np.ndarray) else a

    def _map(self, batch, fn):
        if isinstance(batch, dict):
            return {k: self._map(v, fn) for k, v in batch.items()}
        if isinstance(batch, (tuple, list)):
            return type(batch)(self._map(v, fn) for v in batch)
        if isinstance(batch, (np.ndarray, torch.Tensor)):
            return fn(self._to_tensor(batch))
        return batch

    def _upload(self, batch):
        if not self.cuda:
            out = self._map(batch, lambda t: t)
        else:
            with torch.cuda.stream(self.stream):
                out = self._map(batch, lambda t: (t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True))
        if self.lengths is not None:
            fk, lk = self.lengths
            lens = [int(v) for v in self._to_tensor(batch[lk]).reshape(-1).tolist()]   # host values: no sync
            if self.validate:
                self._check_lengths(self._to_tensor(batch[fk]), lens)
            out[fk]._mmnas_lengths = lens
        return out

    def _check_lengths(self, feat, lens):
        """The region counts against the all-zero-row mask the networks derive from the same features (see CONTRACT)."""
        if feat.is_cuda or feat.dim() != 3 or len(lens) != feat.shape[0]:
            raise ValueError('DevicePrefetcher(lengths=...): %d counts for features of shape %s' % (len(lens), tuple(feat.shape)))
        S = feat.shape[1]
        # numpy on the host view, not torch: a torch CPU op over > 32 K elements opens an OpenMP region over every core of
        # the box (256 logical CPUs on the GPU boxes) -- six of them per batch cost 28 ms a step, measured; numpy's
        # single-threaded gather of two rows per sample costs 0.1 ms
        f = feat.detach().numpy()
        n = np.asarray(lens, dtype=np.int64)
        if self.validate == 'full':
            nz = (f != 0).any(-1)
            want = np.arange(S)[None, :] < n[:, None]
            bad = np.nonzero((nz != want).any(-1))[0].tolist()
        else:      # two gathered rows per sample: row n_b - 1 must be non-zero, row n_b all-zero
            rng = (n >= 0) & (n <= S)
            nc = np.clip(n, 0, S)
            b = np.arange(len(lens))
            last_ok = (nc == 0) | (f[b, np.clip(nc - 1, 0, None)] != 0).any(-1)
            next_ok = (nc == S) | ~(f[b, np.clip(nc, None, S - 1)] != 0).any(-1)
            bad = np.nonzero(~(rng & last_ok & next_ok))[0].tolist()
        if bad:
            raise ValueError('DevicePrefetcher: region counts disagree with the zero-row padding of the features for samples %s '
                             '(count n_b: rows < n_b non-zero, rows >= n_b all-zero)' % bad[:8])

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while True:
            if self.cuda:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_stream(self.stream)
                self._map(nxt, lambda t: t.record_stream(cur) if t.is_cuda else None)
            ready = nxt
            try:
                nxt = self._upload(next(it))
            except StopIteration:
                yield ready
                return
            yield ready

    def __len__(self):
        return len(self.loader)
