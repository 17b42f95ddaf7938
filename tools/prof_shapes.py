"""Aggregate a MMNAS_PROF_DUMP file (one row per bracketed launch: kind,tag,ms,flops,bytes) by tag.

    MMNAS_PROF_DUMP=gpurun_out/shapes.csv python bench.py --steps 5 --no-cpu-baseline
    python tools/prof_shapes.py gpurun_out/shapes.csv 5
"""
import collections
import sys

KINDS = ['gemm', 'mha_fwd', 'mha_bwd', 'rel_fwd', 'rel_bwd', 'rowops', 'lstm', 'small_ops']


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    agg = collections.OrderedDict()
    for line in open(path):
        kind, tag, ms, flops, nbytes = line.rstrip('\n').split(',')
        key = (int(kind), tag)
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(ms)
        a[2] += float(flops)
        a[3] += float(nbytes)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print('%-9s %-62s %7s %9s %9s %8s %8s' % ('kind', 'tag', 'n/step', 'ms/step', 'us/launch', 'TF/s', 'GB/s'))
    for (kind, tag), (n, ms, fl, by) in rows:
        print('%-9s %-62s %7.1f %9.4f %9.1f %8.1f %8.0f' % (KINDS[kind], tag, n / steps, ms / steps, 1e3 * ms / n,
                                                          fl / ms / 1e9 if ms else 0, by / ms / 1e6 if ms else 0))
    print('total ms/step %.3f' % (tot / steps))


if __name__ == '__main__':
    main()
