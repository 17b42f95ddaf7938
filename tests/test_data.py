"""Data path (mmnas_amd/data.py): known-answer tests of the loader restatements (host logic, CPU) and the device
prefetcher (pass-through on CPU; the GPU path is covered in test_data_gpu below)."""
import io

import numpy as np
import pytest
import torch

from tests.util import rel_err  # noqa: F401  (also puts the repo on sys.path)
from mmnas_amd import data


def test_pad_rows_truncates_and_zero_pads():
    a = np.arange(12, dtype=np.float32).reshape(4, 3)
    p = data.pad_rows(a, 6)
    assert p.shape == (6, 3) and np.array_equal(p[:4], a) and not p[4:].any()
    t = data.pad_rows(a, 2)
    assert np.array_equal(t, a[:2])
    assert data.pad_rows(np.zeros((0, 3), np.float32), 2).shape == (2, 3)


def test_bbox_features_known_answer():
    # load_data_vqa.py:266-275: x over the image width (img_shape[1]), y over the height, area fraction
    bbox = np.array([[10., 20., 110., 220.], [0., 0., 640., 480.]], np.float32)
    f = data.bbox_features(bbox, (480, 640))
    exp = np.array([[10 / 640, 20 / 480, 110 / 640, 220 / 480, 100 * 200 / (480 * 640)], [0, 0, 1, 1, 1]], np.float32)
    assert f.dtype == np.float32 and np.allclose(f, exp, rtol=1e-6)


def test_load_frcn_npz_and_collate():
    rs = np.random.RandomState(0)
    samples = []
    for n in (37, 100, 120):
        buf = io.BytesIO()
        np.savez(buf, x=rs.standard_normal((2048, n)).astype(np.float32), bbox=rs.uniform(0, 400, (n, 4)).astype(np.float32),
                 image_h=np.int64(480), image_w=np.int64(640))
        buf.seek(0)
        samples.append(data.load_frcn_npz(buf, pad_size=100))
    s = samples[0]
    assert s['frcn_feat'].shape == (100, 2048) and s['bbox_feat'].shape == (100, 5) and s['bbox'].shape == (100, 4)
    assert s['nobj'] == 37 and not s['frcn_feat'][37:].any() and s['frcn_feat'][:37].any()
    assert samples[2]['nobj'] == 100                      # more boxes than the pad size: truncated
    b = data.collate_regions(samples)
    assert b['frcn_feat'].shape == (3, 100, 2048) and b['nobj'].tolist() == [37, 100, 100] and b['nobj'].dtype == np.int32


def test_prefetcher_is_a_pass_through_on_cpu():
    batches = [({'a': np.full((2, 3), i, np.float32)}, torch.full((4,), float(i)), 'tag%d' % i) for i in range(4)]
    out = list(data.DevicePrefetcher(batches, 'cpu'))
    assert len(out) == 4
    for i, (d, t, tag) in enumerate(out):
        assert isinstance(d['a'], torch.Tensor) and float(d['a'][0, 0]) == i and float(t[0]) == i and tag == 'tag%d' % i
    assert list(data.DevicePrefetcher([], 'cpu')) == []


@pytest.mark.gpu
def test_prefetcher_and_relations_on_the_gpu():
    from oracle import mmnas_oracle as O
    rs = np.random.RandomState(3)
    batches = []
    for i in range(5):
        x1, y1 = rs.uniform(0, 300, (4, 20)), rs.uniform(0, 200, (4, 20))
        bbox = np.stack([x1, y1, x1 + rs.uniform(1, 99, (4, 20)), y1 + rs.uniform(1, 99, (4, 20))], -1).astype(np.float32)
        batches.append({'bbox': bbox, 'nobj': np.array([20, 3, 11, 1], np.int32), 'feat': rs.standard_normal((4, 20, 64)).astype(np.float32)})
    seen = 0
    for ref, got in zip(batches, data.DevicePrefetcher(batches, 'cuda')):
        assert got['feat'].is_cuda and np.array_equal(got['feat'].cpu().numpy(), ref['feat'])
        rel = data.relations_on_device(got['bbox'], got['nobj']).cpu().numpy()
        for b in range(4):
            n = int(ref['nobj'][b])
            exp = np.zeros((20, 20, 4), np.float32)
            exp[:n, :n] = O.relation_embedding(torch.from_numpy(ref['bbox'][b, :n]).double()).numpy()
            assert rel_err(rel[b], exp) < 1e-4
        seen += 1
    assert seen == 5


# ---- the loader functions against the reference's own outputs (tests/golden/loader.npz) ----------------------------
def test_host_functions_vs_reference_loader():
    from tests.golden import cases
    from tests.util import load
    npz = load('loader.npz')
    for i in range(3):
        want = npz['pad|%d|out' % i]
        assert np.array_equal(data.pad_rows(npz['pad|%d|in' % i], want.shape[0]), want)
    for i in range(2):
        out = data.bbox_features(npz['bboxfeat|%d|bbox' % i], tuple(npz['bboxfeat|%d|shape' % i]))
        assert np.array_equal(out, npz['bboxfeat|%d|out' % i])
    tok = {w: i for i, w in enumerate(cases.LOADER_VOCAB)}
    for i, q in enumerate(cases.LOADER_QUESTIONS):
        ix, n = data.tokenize(q, tok, 14)
        assert np.array_equal(ix, npz['sem|%d|ques_ix' % i])
        assert n == npz['sem|%d|out' % i].shape[0]


def test_hard_negative_selection():
    """train_itm.py:349-353 on a small case worked out by hand: per anchor the dataset indices of the best-scoring
    candidates, best first."""
    from mmnas_amd.harness import hard_negative_indices
    scores = torch.tensor([0.1, 0.9, 0.5, 0.3,   0.7, 0.2, 0.8, 0.4])
    neg_idx = torch.tensor([[10, 11, 12, 13], [20, 21, 22, 23]])
    assert hard_negative_indices(scores, neg_idx, 2).tolist() == [[11, 12], [22, 20]]
    assert hard_negative_indices(scores.view(2, 4), neg_idx, 4).tolist() == [[11, 12, 13, 10], [22, 20, 23, 21]]


def test_answer_targets_vs_reference_loader():
    from tests.golden import cases
    from tests.util import load
    npz = load('loader.npz')
    a2i = {a: i for i, a in enumerate(cases.LOADER_ANSWERS)}
    out = data.answer_targets(list(cases.LOADER_ANSWER_SETS), a2i)
    for i in range(len(cases.LOADER_ANSWER_SETS)):
        assert np.array_equal(out[i], npz['ans|%d|out' % i]), i
    up = data.answer_targets([[a.upper() for a in s] for s in cases.LOADER_ANSWER_SETS], a2i, normalize=str.lower)
    assert np.array_equal(up, out)


@pytest.mark.gpu
def test_device_relation_kernels_vs_reference_loader():
    """relation_embedding / semantic_embedding of load_data_vqa.py:7-58 on the GPU, batched and zero-padded as
    DataSet.__getitem__ does (load_data_vqa.py:221-239), against the reference functions' own outputs."""
    from tests.golden import cases
    from tests.util import load
    npz = load('loader.npz')
    # boxes: four samples of 7 / 36 / 100 / 1 objects in one padded batch
    S = 100
    bbox = np.zeros((4, S, 4), np.float32)
    nobj = []
    for i in range(4):
        b = npz['rel|%d|bbox' % i]
        bbox[i, :b.shape[0]] = b
        nobj.append(b.shape[0])
    rel = data.relations_on_device(torch.from_numpy(bbox).cuda(), torch.tensor(nobj, dtype=torch.int32).cuda()).cpu().numpy()
    for i, n in enumerate(nobj):
        want = np.zeros((S, S, 4), np.float32)
        want[:n, :n] = npz['rel|%d|out' % i]
        assert np.allclose(rel[i], want, rtol=2e-5, atol=2e-5), i
    # questions
    tok = {w: i for i, w in enumerate(cases.LOADER_VOCAB)}
    ixs, ns = zip(*[data.tokenize(q, tok, 14) for q in cases.LOADER_QUESTIONS])
    emb = torch.from_numpy(npz['sem|emb']).cuda()
    out = data.semantic_relations_on_device(torch.from_numpy(np.stack(ixs)).cuda(), torch.tensor(ns, dtype=torch.int32).cuda(), emb)
    out = out.cpu().numpy()
    for i, n in enumerate(ns):
        want = np.zeros((14, 14, 3), np.float32)
        want[:n, :n] = npz['sem|%d|out' % i]
        assert np.allclose(out[i], want, rtol=2e-5, atol=2e-5), i


def _feat(lens, S=5, F=4):
    import torch
    f = torch.zeros(len(lens), S, F)
    for b, n in enumerate(lens):
        f[b, :n] = 1.0 + b
    return f


def test_prefetcher_attaches_the_region_counts_to_the_features():
    """DevicePrefetcher(lengths=...): the loaders' region counts ride on the features tensor (CPU pass-through here)."""
    import torch
    from mmnas_amd.data import DevicePrefetcher
    batches = [(_feat([5, 2, 4]), torch.tensor([5, 2, 4], dtype=torch.int32)), {'f': _feat([1, 5]), 'n': np.array([1, 5])}]
    out = list(DevicePrefetcher(batches[:1], 'cpu', lengths=(0, 1)))
    assert out[0][0]._mmnas_lengths == [5, 2, 4]
    out = list(DevicePrefetcher(batches[1:], 'cpu', lengths=('f', 'n')))
    assert out[0]['f']._mmnas_lengths == [1, 5]


def test_prefetcher_checks_the_counts_against_the_zero_row_padding():
    """the ragged stream trusts the counts the pipeline attaches, while the masks come from the all-zero feature
    rows -- the two must agree.  Checked on the host copy before the upload: boundary rows by default, every row with
    validate='full'; a disagreement raises instead of silently changing logits."""
    import pytest
    import torch
    from mmnas_amd.data import DevicePrefetcher
    good = _feat([5, 2, 4])
    for lens in ([5, 3, 4], [5, 1, 4], [4, 2, 4]):            # count too large / too small for the zero-row mask
        with pytest.raises(ValueError, match='region counts disagree'):
            list(DevicePrefetcher([(good, torch.tensor(lens))], 'cpu', lengths=(0, 1)))
    hole = _feat([5, 4, 4])
    hole[1, 1] = 0                                              # an all-zero row INSIDE the prefix: only the full check sees it
    assert list(DevicePrefetcher([(hole, torch.tensor([5, 4, 4]))], 'cpu', lengths=(0, 1)))
    with pytest.raises(ValueError, match='region counts disagree'):
        list(DevicePrefetcher([(hole, torch.tensor([5, 4, 4]))], 'cpu', lengths=(0, 1), validate='full'))
    assert list(DevicePrefetcher([(hole, torch.tensor([5, 3, 4]))], 'cpu', lengths=(0, 1), validate=None))   # unchecked on request
    with pytest.raises(ValueError, match='counts for features'):
        list(DevicePrefetcher([(good, torch.tensor([5, 2]))], 'cpu', lengths=(0, 1)))
