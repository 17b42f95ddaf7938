"""H2D copies of a streamed-input run against the kernel timeline (rocprofv3 --kernel-trace --memory-copy-trace CSVs).

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr -o t -- python3 bench.py --workload search_vqa_stream ...
    python tools/copy_overlap.py /tmp/tr > profiles/r03_timeline_search_vqa_stream.txt

For every host-to-device copy of the last steady-state steps: start offset inside its step, duration, bytes, GB/s, and the
share of its duration during which a compute kernel was running (1.00 = completely hidden behind compute)."""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'onehot_rows'
    kf = sorted(glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True))[0]
    cf = sorted(glob.glob(os.path.join(root, '**', '*memory_copy_trace.csv'), recursive=True))[0]
    ker = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kf)))
    cps = []
    for r in csv.DictReader(open(cf)):
        d = r.get('Direction') or r.get('Name') or ''
        nbytes = int(r.get('Bytes') or r.get('Size') or 0) if (r.get('Bytes') or r.get('Size')) else 0
        cps.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), d, nbytes))
    cps.sort()
    starts = [s for s, e, n in ker if marker in n]
    if len(starts) < 4:
        print('fewer than 4 steps found')
        return
    lo, hi = starts[-4], starts[-1]
    print('three steady-state steps: %.1f us each on average; copies inside them:' % ((hi - lo) / 3e3))
    print('(hidden = share of the copy during which a compute kernel was running; slack = how long before the first kernel of the')
    print(' NEXT step the copy had finished -- a positive slack means the copy was not what that step waited for.  Under the')
    print(' profiler every step has a ~0.3 ms host-side hole near its start, which is where the unhidden part of a copy falls.)')
    print('%10s %9s %12s %8s %8s %9s  %s' % ('at us', 'dur us', 'bytes', 'GB/s', 'hidden', 'slack us', 'direction'))
    tot = hid = 0.0
    for s, e, d, nb in cps:
        if not (lo <= s < hi) or 'HOST_TO_DEVICE' not in d.upper().replace(' ', '_') and 'H2D' not in d.upper():
            continue
        ov = 0
        for ks, ke, _ in ker:
            if ke <= s:
                continue
            if ks >= e:
                break
            ov += min(e, ke) - max(s, ks)
        dur = max(e - s, 1)
        step0 = max(x for x in starts if x <= s)
        nxt = [x for x in starts if x > s]
        slack = (nxt[0] - e) / 1e3 if nxt else float('nan')
        print('%10.1f %9.1f %12d %8.1f %8.2f %9.1f  %s' % ((s - step0) / 1e3, dur / 1e3, nb, nb / dur, min(ov / dur, 1.0), slack, d))
        tot += dur
        hid += min(ov, dur)
    if tot:
        print('copy time %.1f us per step, %.2f of it behind compute kernels' % (tot / 3e3, hid / tot))


if __name__ == '__main__':
    main()
