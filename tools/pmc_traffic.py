"""HBM-side traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected separately,
as MI355X_MICROARCH.md prescribes: they do not fit one pass).

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> out.json

Per kernel name: launches, mean FETCH_SIZE and WRITE_SIZE (KB, as rocprofv3 reports them), and
traffic_bytes = 2 * FETCH + WRITE -- on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at
64 bytes, so it is doubled (same guide, HBM section); WRITE_SIZE is exact for 16-byte stores and float atomics.
"""
import collections
import glob
import json
import os
import sqlite3
import sys


def collect(d, counter):
    out = collections.defaultdict(lambda: [0, 0.0])
    db = glob.glob(os.path.join(d, '**', '*_results.db'), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if 'pmc_event' in t][0]
    info = [t for t in tabs if 'info_pmc' in t][0]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    q = (f"select s.kernel_name, d.id, sum(e.value) from {pmc} e join {info} i on e.pmc_id=i.id "
         f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id where i.name=? group by d.id")
    for name, _did, val in cur.execute(q, (counter,)):
        key = name.split('(')[0].replace('void ', '')[:90]
        out[key][0] += 1
        out[key][1] += val
    return out


def main():
    f = collect(sys.argv[1], 'FETCH_SIZE')
    w = collect(sys.argv[2], 'WRITE_SIZE')
    res = {}
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        fk = sf / nf if nf else 0.0
        wk = sw / nw if nw else 0.0
        res[k] = {'launches': max(nf, nw), 'fetch_kb_per_launch': fk, 'write_kb_per_launch': wk,
                  'traffic_bytes_per_launch': (2.0 * fk + wk) * 1024.0}
    json.dump(res, open(sys.argv[3], 'w'), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]['traffic_bytes_per_launch'] * kv[1]['launches'])[:12]:
        print('%-70s n=%4d  fetch %9.1f KB  write %9.1f KB  traffic %8.2f MB/launch'
              % (k[:70], v['launches'], v['fetch_kb_per_launch'], v['write_kb_per_launch'], v['traffic_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
