"""Turn a rocprofv3 --kernel-trace --stats output directory into the committed summary under profiles/.

    python tools/summarize_prof.py gpurun_out/prof2 profiles/r01_train_vqa auto "command line that was profiled"

The step count is taken FROM THE TRACE (`auto`: the calls of lstm_seq_fwd_kernel -- every step of every workload runs the
question LSTM forward exactly once; `auto/3` for the ITM triplet step's three forwards); a literal count is still accepted
but is checked against the trace and the trace wins.  (Round 4 passed a literal 23 for a command that ran 26 steps: every
per-step column of the r04 .md files is 13 % too high.)

Writes <out>_kernel_stats.csv (verbatim copy of rocprofv3's per-kernel stats) and <out>.md (top
kernels per step, GEMM durations grouped by launch geometry)."""
import collections
import csv
import glob
import os
import shutil
import sys


def steps_from_trace(rows, arg):
    """-> (steps, how it was found).  The marker is the persistent question LSTM's forward launch: one per step."""
    per = 1
    if arg.startswith('auto/'):
        per = int(arg.split('/')[1])
    marker = [int(r['Calls']) for r in rows if 'lstm_seq_fwd_kernel' in r['Name']]
    counted = sum(marker) // per if marker else None
    if arg.startswith('auto'):
        if counted is None:
            raise SystemExit('summarize_prof: no lstm_seq_fwd_kernel launch in the trace: pass the step count')
        return counted, 'counted in the trace: %d lstm_seq_fwd_kernel launches / %d per step' % (sum(marker), per)
    if counted is not None and counted != int(arg):
        sys.stderr.write('summarize_prof: %s steps given, the trace holds %d: using the trace\n' % (arg, counted))
        return counted, 'counted in the trace (the command line said %s)' % arg
    return int(arg), 'as given'


def main():
    src, out, steps_arg = sys.argv[1], sys.argv[2], sys.argv[3]
    cmd = sys.argv[4] if len(sys.argv) > 4 else ''
    stats = glob.glob(os.path.join(src, '**', '*_kernel_stats.csv'), recursive=True)[0]
    trace = glob.glob(os.path.join(src, '**', '*_kernel_trace.csv'), recursive=True)[0]
    shutil.copy(stats, out + '_kernel_stats.csv')
    rows = list(csv.DictReader(open(stats)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    steps, how = steps_from_trace(rows, steps_arg)
    lines = ['# rocprofv3 --kernel-trace --stats summary', '', '`%s`' % cmd, '',
             'steps profiled (warm-up included): %d (%s); device-busy time %.2f ms/step' % (steps, how, tot / steps / 1e6), '',
             '| ms/step | launches/step | avg us | % | kernel |', '|---|---|---|---|---|']
    # Which PASS of the command a kernel's launches belong to (the per-step averages above are over ALL steps of the
    # command -- warm-up, timed, empty-queue and roofline-pass steps; `__amd_rocclr_copyBuffer` is not in a timed step at all).
    # The passes are given as step counts, SUMMARIZE_PASSES="3,10,3,10" = warm-up, timed, empty-queue, roofline pass; steps are
    # delimited in the trace by the launches of the per-step marker kernel (lstm_seq_fwd_kernel).
    per_pass = {}
    passes = [int(x) for x in os.environ.get('SUMMARIZE_PASSES', '').split(',') if x]
    if passes and sum(passes) == steps:
        tr = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Start_Timestamp']))
        marks = [int(r['Start_Timestamp']) for r in tr if 'lstm_seq_fwd_kernel' in r['Kernel_Name']]
        bounds, acc = [], 0
        for n in passes:
            bounds.append((acc, acc + n))
            acc += n
        timed_lo, timed_hi = bounds[1]
        t0 = marks[timed_lo]
        t1 = marks[timed_hi] if timed_hi < len(marks) else None
        for r in tr:
            ts = int(r['Start_Timestamp'])
            if ts >= t0 and (t1 is None or ts < t1):
                k = r['Kernel_Name']
                per_pass[k] = per_pass.get(k, 0) + 1
        lines[-2] = '| ms/step | launches/step | launches/step in the %d TIMED steps | avg us | %% | kernel |' % passes[1]
        lines[-1] = '|---|---|---|---|---|---|'
    for r in rows[:25]:
        name = r['Name'].replace('void ', '').replace('|', '/')[:110]
        if per_pass:
            lines.append('| %.3f | %.1f | %.1f | %.1f | %.1f | `%s` |' % (float(r['TotalDurationNs']) / steps / 1e6, int(r['Calls']) / steps,
                                                                        per_pass.get(r['Name'], 0) / passes[1], float(r['AverageNs']) / 1e3,
                                                                        float(r['Percentage']), name))
        else:
            lines.append('| %.3f | %.1f | %.1f | %.1f | `%s` |' % (float(r['TotalDurationNs']) / steps / 1e6, int(r['Calls']) / steps,
                                                                  float(r['AverageNs']) / 1e3, float(r['Percentage']), name))
    if per_pass:
        cb = sum(v for k, v in per_pass.items() if 'copyBuffer' in k)
        lines += ['', '(`__amd_rocclr_copyBuffer`: %.1f launches per TIMED step -- the launches in the first column come from the roofline '
                  'pass, whose per-launch HIP events the runtime resolves with buffer copies, and from the warm-up; they are not part of '
                  'the step `value` times.)' % (cb / passes[1])]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        n = r['Kernel_Name']
        if 'gemm_kernel' not in n and 'gemm_pair_kernel' not in n:
            continue
        which = 'gemm_pair_kernel' if 'gemm_pair_kernel' in n else 'gemm_kernel'
        key = (('pair' if which == 'gemm_pair_kernel' else '') + n.split(which)[1].split('(')[0], int(r['Grid_Size_X']) // 256,
               int(r['Grid_Size_Z']))
        agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    gt = sum(sum(v) for v in agg.values())
    lines += ['', '## gemm_kernel / gemm_pair_kernel by launch geometry (%.2f ms/step, %.1f launches/step, average %.1f us)'
              % (gt / steps / 1e3, sum(len(v) for v in agg.values()) / steps, gt / max(1, sum(len(v) for v in agg.values()))), '',
              '| template <BM,BN,A k-contig,B k-contig,fast,split parts> (pair<BM,BN,parts>: NN + TN sections) | workgroups | grid.z | launches/step | avg us | ms/step |',
              '|---|---|---|---|---|---|']
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:16]:
        lines.append('| `%s` | %d | %d | %.1f | %.1f | %.2f |' % (k[0], k[1], k[2], len(v) / steps, sum(v) / len(v), sum(v) / steps / 1e3))
    open(out + '.md', 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:14]))


if __name__ == '__main__':
    main()
