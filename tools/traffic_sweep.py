"""Does the fabric traffic of the GEMM matter?  Same product, different tile orders / XCD mappings -> different
FETCH_SIZE, duration compared (addresses VERDICT r01 weak #4: "no run shows kernel time unchanged with FETCH halved").

    GEMM_PMC_SWEEP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/sweep -o t -- python3 tools/gemm_pmc.py
    python tools/traffic_sweep.py /tmp/sweep > profiles/r02_gemm_traffic_sweep.txt

tools/gemm_pmc.py (sweep mode) launches, for xcd-remap in (1, 0) and gm (row-panels per tile-order block) in (1, 8, 16),
4 repetitions each of NT 6400x2048x512 and NT 6400x512x512; this groups the dispatches in that order."""
import glob
import os
import sqlite3
import sys


def main():
    db = glob.glob(os.path.join(sys.argv[1], '**', '*_results.db'), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if 'pmc_event' in t][0]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = cur.execute(f"select d.id, s.kernel_name, d.start, d.end, (select sum(e.value) from {pmc} e where e.event_id=d.event_id) "
                       f"from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
    g = [(r[3] - r[2], r[4] or 0.0) for r in rows if 'gemm_kernel' in r[1]]
    variants = [(x, gm, shp) for x in (1, 0) for gm in (1, 8, 16) for shp in ('6400x2048x512', '6400x512x512')]
    print('FETCH_SIZE is doubled (gfx950 counts 128-byte requests at 64 bytes); algorithmic operand bytes: 6400x2048x512: 17.3 MB, 6400x512x512: 14.2 MB\n')
    print('%-14s %-4s %-3s | %10s | %12s | %s' % ('shape', 'xcd', 'gm', 'time us', '2*FETCH MB', 'vs gm=8,xcd=1'))
    base = {}
    res = []
    for i, (x, gm, shp) in enumerate(variants):
        grp = g[4 * i:4 * i + 4]
        if len(grp) < 4:
            break
        t = min(d for d, _ in grp[1:]) / 1e3           # first repetition is cold: skip it
        f = sum(v for _, v in grp[1:]) / 3 * 2 * 1024 / 1e6   # KB -> MB, doubled
        res.append((shp, x, gm, t, f))
        if x == 1 and gm == 8:
            base[shp] = (t, f)
    for shp, x, gm, t, f in res:
        bt, bf = base.get(shp, (t, f))
        print('%-14s %-4d %-3d | %10.1f | %12.1f | time x%.2f  fetch x%.2f' % (shp, x, gm, t, f, t / bt, f / bf))


if __name__ == '__main__':
    main()
