"""Summarise a rocprofv3 --kernel-trace results .db: consecutive launches of the same kernel/grid are grouped,
mean duration and mean gap to the next launch printed (tuning aid for A/B microbenchmarks)."""
import itertools
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_z, d.group_segment_size "
                       f"from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    minrun = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for key, grp in itertools.groupby(rows, key=lambda r: (r[0][:64], r[3], r[4], r[5])):
        g = list(grp)
        if len(g) < minrun:
            continue
        durs = [(r[2] - r[1]) / 1e3 for r in g]
        gaps = [(g[i + 1][1] - g[i][2]) / 1e3 for i in range(len(g) - 1)]
        print('%-66s grid %7d x%d lds %6d  n=%3d  dur %7.1f us  gap %5.1f us' % (key[0], key[1], key[2], key[3], len(g),
              sum(durs) / len(durs), sum(gaps) / max(1, len(gaps))))


if __name__ == '__main__':
    main()
