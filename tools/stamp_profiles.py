"""Stamp the PMC-derived JSON files of a round with the tree they were measured on (run in the build container, where
.git exists, right after a GPU call has merged them back):

    python tools/stamp_profiles.py profiles/r03_traffic_search_vqa.json [...]

Adds "_meta": {"commit": <HEAD the measured snapshot was taken from, + "-dirty" if the tree had changes>,
"library_md5": md5 of mmnas_amd/lib/libmmnas_hip.so}.  bench.py copies it into roofline.traffic_pmc so that a stale
counter file shows in the benchmark line itself.
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if os.environ.get('MMNAS_COMMIT'):     # on the GPU box (no .git there): the commit the snapshot was taken from, passed in
        head, dirty = os.environ['MMNAS_COMMIT'], ''
    else:
        head = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        dirty = subprocess.run(['git', 'status', '--porcelain', '--', 'mmnas_amd', 'bench.py'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    lib = os.path.join(ROOT, 'mmnas_amd', 'lib', 'libmmnas_hip.so')
    md5 = hashlib.md5(open(lib, 'rb').read()).hexdigest() if os.path.exists(lib) else None
    for path in sys.argv[1:]:
        d = json.load(open(path))
        d['_meta'] = {'commit': head + ('-dirty' if dirty else ''), 'library_md5': md5}
        json.dump(d, open(path, 'w'), indent=1)
        print(path, d['_meta'])


if __name__ == '__main__':
    main()
