"""Per-kernel hardware counters from `rocprofv3 --pmc` passes of the benchmark -> one JSON for profiles/.

    python tools/pmc_counters.py out.json <pass dir> [<pass dir> ...]

Each pass directory holds one rocprofv3 run with a few counters (they do not all fit one pass; --kernel-trace only,
never a sys/hip trace next to --pmc).  Per kernel name the tool reports launches and the mean per-launch value of every
counter it finds, plus
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)  (the busy counter runs per SIMD, four per CU: the
                     share of a busy CU's SIMD-cycles with the MFMA pipe busy; both summed over the chip by rocprofv3)
    lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE  (when both are present)
and a `classes` section that adds the kernels up into the bench's classes (gemm, mha_fwd, mha_bwd, rel, rowops, lstm).
"""
import collections
import glob
import json
import os
import sqlite3
import sys

CLASSES = (('gemm', ('gemm_kernel', 'gemm_pair_kernel')), ('mha_fwd', ('mha_fwd_kernel',)),
           ('mha_bwd', ('mha_bwd_fused_kernel', 'mha_bwd_q_kernel', 'mha_bwd_kv_kernel')),
           ('rel_fwd', ('rel_fused_fwd_kernel', 'rel_bias_fwd_kernel')), ('rel_bwd', ('rel_fused_bwd_kernel', 'rel_bias_bwd_kernel', 'rel_fused_reduce')),
           ('rowops', ('ln_fwd_kernel', 'ln_bwd_kernel', 'colsum_kernel', 'drop_add_kernel', 'mixed_sum')),
           ('lstm', ('lstm_seq_fwd_kernel', 'lstm_seq_bwd_kernel')))


def collect(d):
    """{kernel: {counter: [launches, sum]}} of one pass."""
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    db = glob.glob(os.path.join(d, '**', '*_results.db'), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if 'pmc_event' in t][0]
    info = [t for t in tabs if 'info_pmc' in t][0]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    q = (f"select s.kernel_name, i.name, d.id, sum(e.value) from {pmc} e join {info} i on e.pmc_id=i.id "
         f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by d.id, i.name")
    for name, counter, _did, val in cur.execute(q):
        key = name.split('(')[0].replace('void ', '')[:90]
        c = out[key][counter]
        c[0] += 1
        c[1] += val
    return out


def main():
    dst, passes = sys.argv[1], sys.argv[2:]
    merged = collections.defaultdict(dict)
    for d in passes:
        for k, cs in collect(d).items():
            for c, (n, s) in cs.items():
                merged[k][c] = (n, s)
    res = {'kernels': {}, 'classes': {}}
    for k, cs in sorted(merged.items()):
        e = {'launches': max(n for n, _ in cs.values())}
        for c, (n, s) in cs.items():
            e[c] = s / n
        res['kernels'][k] = e
    def derive(e):
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and e.get('SQ_BUSY_CU_CYCLES'):
            e['mfma_busy_frac'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * e['SQ_BUSY_CU_CYCLES'])
        if 'SQ_LDS_BANK_CONFLICT' in e and e.get('SQ_LDS_IDX_ACTIVE'):
            e['lds_conflict_frac'] = e['SQ_LDS_BANK_CONFLICT'] / e['SQ_LDS_IDX_ACTIVE']
    for e in res['kernels'].values():
        derive(e)
    for cname, pats in CLASSES:
        tot = collections.defaultdict(float)
        launches = 0
        for k, cs in merged.items():
            if any(p in k for p in pats):
                launches += max(n for n, _ in cs.values())
                for c, (n, s) in cs.items():
                    tot[c] += s
        if launches:
            e = {'launches': launches}
            e.update({c + '_total': v for c, v in tot.items()})
            d = {c: v for c, v in tot.items()}
            derive(d)
            e.update({k: v for k, v in d.items() if k.endswith('_frac')})
            res['classes'][cname] = e
    json.dump(res, open(dst, 'w'), indent=1)
    for cname, e in res['classes'].items():
        print('%-8s launches %5d  %s' % (cname, e['launches'], '  '.join('%s %.3f' % (k, v) for k, v in e.items() if k.endswith('_frac'))))


if __name__ == '__main__':
    main()
