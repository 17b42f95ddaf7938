"""Tuning aid: for every kernel of a hipcc -save-temps assembly file, report scratch (spill) instructions that sit in a
basic-block range containing MFMA instructions between a label and its backward branch (i.e. inside a hot loop).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -c gemm.hip -o /tmp/gemm.o -save-temps=obj
    python tools/asm_loop_spills.py /tmp/gemm-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import subprocess
import sys


def main():
    lines = open(sys.argv[1]).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
    for si, s in enumerate(starts):
        e = starts[si + 1] if si + 1 < len(starts) else len(lines)
        name = lines[s].split(':')[0]
        body = lines[s:e]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\w+):', l)] if m}
        hot = []
        for i, l in enumerate(body):
            m = re.search(r's_cbranch_\w+\s+(\.LBB\w+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:      # backward branch: a loop
                lo, hi = labels[m.group(1)], i
                seg = body[lo:hi]
                nm = sum('v_mfma' in x for x in seg)
                ns = sum('scratch_' in x for x in seg)
                if nm:
                    hot.append((lo, hi, nm, ns))
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        total = sum('scratch_' in x for x in body)
        print('%-100s scratch ops %3d | loops with MFMA: %s' % (dem[:100], total, ', '.join('mfma %d scratch %d' % (a[2], a[3]) for a in hot)))


if __name__ == '__main__':
    main()
