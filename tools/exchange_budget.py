"""Bytes, completion points and predicted exposed communication of the data-parallel gradient exchange.

    python tools/exchange_budget.py            # CPU only: builds the bench's nets and reducers, no process group

For the supernet weight step (dp.SupernetReducer, 3 buckets) and the fixed-architecture step (dp.GradReducer, 64 MB
buckets): what each bucket carries, which backbone operator issues its last gradient (= where its pack + all-reduce can
start: the mark the chain executor records), and -- from the measured single-GPU backward timeline -- how much backward is
left behind that point to hide the collective.  xGMI model: 8 GPUs fully connected, 7 links x 153 GB/s per
GPU.  A ring all-reduce moves 2 (N-1)/N x bytes over ONE link per direction; a direct reduce-scatter + all-gather over
N-1 links moves 2 x bytes / N per link.  Both are printed; RCCL picks between them by size.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LINK = 153e9


def t_ring(nbytes, n):
    return 2.0 * (n - 1) / n * nbytes / LINK


def t_direct(nbytes, n):
    return 2.0 * nbytes / n / LINK


def main():
    from mmnas_amd import dp
    from mmnas_amd.model import mixed
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas.model.full_vqa import Net_Full
    from mmnas.model.mixed import MixedOp
    emb = np.zeros((bench.VOCAB, 300), np.float32)
    init = {'token_size': bench.VOCAB, 'ans_size': bench.ANS, 'pretrained_emb': emb}
    # measured on one MI355X (profiles/r04_bench.json): backward of the supernet step ~ 3.6 ms of the 5.4 ms step, of which
    # the backbone ~ 3.0 ms; training step: backward ~ 7.7 of 11.5 ms
    out = {}
    # round 6: the step times of this round's build (`--search-ms` / `--train-ms`, defaults = profiles/r06 bench on the box of
    # the refresh); backward / backbone shares are the measured proportions of the round-3 timelines (backward 0.667 of the
    # supernet step, the backbone 0.833 of it; 0.67 / 0.91 for the training step) -- a MODEL of where the marks fall, not a
    # multi-GPU measurement: no curve was measured (no multi-GPU box is reachable from the build container)
    def arg(flag, dflt):
        return float(sys.argv[sys.argv.index(flag) + 1]) if flag in sys.argv else dflt
    s_ms, t_ms = arg('--search-ms', 4.45), arg('--train-ms', 10.4)
    tag = sys.argv[sys.argv.index('--tag') + 1] if '--tag' in sys.argv else 'r06'
    for name, bwd_ms, backbone_ms in (('search_vqa', 0.667 * s_ms, 0.667 * 0.833 * s_ms), ('train_vqa', 0.67 * t_ms, 0.67 * 0.91 * t_ms)):
        if name == 'search_vqa':
            cfg = bench.make_cfg('search')
            net = Net_Search(cfg, init)
            red = dp.SupernetReducer(net, n_buckets=3)
            mixed.seed_arch_sampler(888)
            MixedOp.MODE = None
            net.reset_binary_gates()
            red.begin_weight_step()
            mops = net.redundant_modules
            rows = []
            emb_bytes = 4 * net.embedding.weight.numel()
            for b in range(red.n_buckets):
                static = sum(n for _, n in red.bucket_static[b])
                if 0 in red.bucket_static_params[b]:
                    static -= net.embedding.weight.numel()      # (with RCCL the table travels as rows: dp.RowExchange)
                cand = sum(red.cand_span[k][mops[k].active_index[0]][1] for k in red.bucket_nodes[b])
                nodes = red.bucket_nodes[b]
                # operators run in reverse node order in backward; the bucket is complete behind its smallest node index
                last = min(nodes) if nodes else None
                frac_done = 1.0 if last is None else (30 - last) / 30.0
                has_stem = b == red.n_buckets - 1      # the stem's runs always sit in the last bucket: it completes with backward
                rows.append(dict(bucket=b, bytes=4 * (static + cand), nodes=(min(nodes), max(nodes)) if nodes else None,
                                 static_bytes=4 * static, backbone_fraction_done_at_mark=None if has_stem else frac_done))
            out[name] = dict(rows=rows, embedding_dense_bytes=emb_bytes,
                             embedding_row_bytes_per_rank=bench.B_DEFAULT * bench.SX * (300 * 4 + 8))
        else:
            cfg = bench.make_cfg('train')
            net = Net_Full(cfg, init)
            red = dp.GradReducer(list(net.parameters()))
            names = {id(p): k for k, p in net.named_parameters()}
            rows = []
            total = sum(hi - lo for lo, hi, _ in red.buckets)
            acc = 0
            embn = net.embedding.weight.numel()
            total -= embn
            for b, (lo, hi, idxs) in enumerate(red.buckets):
                if 0 in idxs:            # (with RCCL the embedding table travels as rows and is in no bucket: dp.RowExchange)
                    lo += embn
                    idxs = [i for i in idxs if i != 0]
                acc += hi - lo
                rows.append(dict(bucket=b, bytes=4 * (hi - lo), first=names[id(red.fg.params[idxs[0]])], last=names[id(red.fg.params[idxs[-1]])],
                                 cumulative_fraction=acc / total))
            out[name] = dict(rows=rows, embedding_dense_bytes=4 * net.embedding.weight.numel())
        print('== %s (backward %.1f ms, backbone part %.1f ms on one GPU)' % (name, bwd_ms, backbone_ms))
        for r in out[name]['rows']:
            nb = r['bytes']
            if name == 'search_vqa':
                fd = r['backbone_fraction_done_at_mark']
                left = bwd_ms - backbone_ms * fd if fd is not None else 0.0      # backward still to run behind the mark
                where = 'nodes %s, mark behind node %d (%.0f %% of the backbone backward done)' % (r['nodes'], r['nodes'][0], 100 * fd) if fd is not None else 'nodes %s + stem: complete at the END of backward' % (r['nodes'],)
            else:
                left = max(0.0, bwd_ms * (1.0 - r['cumulative_fraction']))
                where = '%s .. %s' % (r['first'], r['last'])
            line = 'bucket %d: %6.2f MB  %s; backward left to hide it: %.2f ms' % (r['bucket'], nb / 1e6, where, left)
            for n in (2, 4, 8):
                tr, td = 1e3 * t_ring(nb, n), 1e3 * t_direct(nb, n)
                line += ' | N=%d ring %.3f / direct %.3f ms (exposed %.3f / %.3f)' % (n, tr, td, max(0.0, tr - left), max(0.0, td - left))
            print(line)
        print('embedding table: %.1f MB dense (what DDP all-reduces); as rows: %.2f MB per rank all-gathered' % (
            out[name]['embedding_dense_bytes'] / 1e6, bench.B_DEFAULT * bench.SX * (300 * 4 + 8) / 1e6))
        for n in (2, 4, 8):
            rowb = bench.B_DEFAULT * bench.SX * (300 * 4 + 8)
            print('  N=%d: dense ring all-reduce %.3f ms (exposed: the table completes last) vs row all-gather %.3f ms' % (
                n, 1e3 * t_ring(out[name]['embedding_dense_bytes'], n), 1e3 * (n - 1) * rowb / LINK))
    out['_model'] = dict(search_step_ms=s_ms, train_step_ms=t_ms, link_GBps=LINK / 1e9, note='predicted from one-GPU timings; never measured on more than one GPU')
    json.dump(out, open(os.path.join(ROOT, 'profiles', tag + '_exchange_budget.json'), 'w'), indent=1, default=str)


if __name__ == '__main__':
    main()
