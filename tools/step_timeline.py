"""Kernel timeline of ONE steady-state step from a rocprofv3 --kernel-trace CSV (tuning aid).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python3 bench.py --workload search_vqa --steps 6 --warmup 3 --no-cpu-baseline --no-prof
    python tools/step_timeline.py /tmp/tr [marker] [--list] > gpurun_out/timeline.txt

A step starts at each launch whose name contains `marker` (default: onehot_rows, the supernet's gate write; use
row_is_zero for train_vqa); the second-to-last complete step is reported: span, busy time, per-kernel totals, the
largest gaps, and with --list every launch in order (start offset, duration, gap to the previous end)."""
import csv
import glob
import os
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    root = args[0]
    marker = args[1] if len(args) > 1 else 'onehot_rows'
    f = root if root.endswith('.csv') else sorted(glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        # (queue / stream id in the name column's tail when the trace has one: launches of a communication stream show as q1, q2 ...)
        q = r.get('Queue_Id') or r.get('Stream_Id') or ''
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1),
                     r['Kernel_Name'] if not q else r['Kernel_Name'].split('(')[0] + ' @q' + str(q)))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if marker in r[3]]
    if len(starts) < 3:
        print('fewer than 3 steps found for marker', marker)
        return
    i0, i1 = starts[-3], starts[-2]
    step = rows[i0:i1]
    t0 = step[0][0]
    span = (rows[i1][0] - t0) / 1e3
    busy = sum(e - s for s, e, _, _ in step) / 1e3
    print('step: %d launches, span %.1f us, busy %.1f us, idle %.1f us' % (len(step), span, busy, span - busy))
    agg = {}
    prev_end = t0
    gaps = []
    for s, e, g, name in step:
        short = name.split('(')[0].replace('void ', '')[:70]
        a = agg.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        gap = (s - prev_end) / 1e3
        if gap > 6:
            gaps.append((gap, (s - t0) / 1e3, short))
        prev_end = max(prev_end, e)
    print('\n%-72s %5s %9s %8s' % ('kernel', 'n', 'total us', 'avg us'))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%-72s %5d %9.1f %8.1f' % (k, n, t, t / n))
    print('\ngaps > 6 us (gap, at, before kernel): total %.1f us in %d gaps' % (sum(g[0] for g in gaps), len(gaps)))
    for g in sorted(gaps, reverse=True)[:25]:
        print('  %7.1f us at %8.1f  %s' % g)
    if '--list' in sys.argv:
        print()
        prev_end = t0
        for s, e, g, name in step:
            print('%9.1f %8.1f gap %6.1f  wg%-6d %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, g, name.split('(')[0].replace('void ', '')[:80]))
            prev_end = max(prev_end, e)


if __name__ == '__main__':
    main()
