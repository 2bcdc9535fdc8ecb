"""HBM traffic per launch of the hot kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE each in a run of
its own, as the MI355X guide prescribes), with the gfx950 correction: FETCH_SIZE tallies the 128-byte requests of wide
coalesced reads at 64 bytes, so read bytes = 2 x FETCH_SIZE; both counters are in KiB.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_...csv> <out.json> [--head GIT_HEAD] [--command "..."]
                      [--per-kernel out.csv]
The JSON carries the sha256 of the kernel sources it was measured on (vit_amd.srchash); bench.py reports the numbers
only while that hash still matches the tree it runs from."""
import argparse
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd.srchash import kernel_source_hash  # noqa: E402


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        tot[r['Kernel_Name']] += float(r['Counter_Value'])
        cnt[r['Kernel_Name']] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def short(name):
    """The name bench.py's kernel timer uses for a GEMM instance; the bare kernel name for the other vitk kernels."""
    if 'gemm256_kernel' in name:
        tc = 'f32out' if ('IfLb' in name or '<float' in name) else 'bf16out'
        if 'Lb0ELb0' in name or 'false, false' in name:
            lay = 'NT'
        elif 'Lb0ELb1' in name or 'false, true' in name:
            lay = 'NN'
        elif 'Lb1ELb1' in name or 'true, true' in name:
            lay = 'TN'
        else:
            lay = 'TT'
        return f'gemm_{lay}_bf16' + ('' if tc == 'bf16out' or lay == 'TN' else '_f32out')
    m = re.search(r'vitk::(\w+)', name) or re.search(r'_ZN4vitk\d+([A-Za-z_0-9]+?_kernel)', name)
    return m.group(1) if m else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch')
    ap.add_argument('write')
    ap.add_argument('out')
    ap.add_argument('--head', default='unknown')
    ap.add_argument('--command', default='')
    ap.add_argument('--per-kernel', default='')
    a = ap.parse_args()
    fetch = per_kernel(a.fetch, 'FETCH_SIZE')
    write = per_kernel(a.write, 'WRITE_SIZE')
    kernels = {}
    for k, (fv, n) in fetch.items():
        s = short(k)
        if s is None or k not in write:
            continue
        wv = write[k][0]
        e = kernels.setdefault(s, {'launches_sampled': 0, 'fetch_kib': 0.0, 'write_kib': 0.0})
        e['launches_sampled'] += n
        e['fetch_kib'] += fv * n
        e['write_kib'] += wv * n
    for s, e in kernels.items():
        n = e['launches_sampled']
        e['fetch_size_kib_per_launch'] = e.pop('fetch_kib') / n
        e['write_size_kib_per_launch'] = e.pop('write_kib') / n
        e['hbm_bytes_per_launch'] = (2.0 * e['fetch_size_kib_per_launch'] + e['write_size_kib_per_launch']) * 1024.0
    out = {'git_head': a.head, 'kernel_source_sha256': kernel_source_hash(), 'command': a.command,
           'formula': '(2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes (gfx950: FETCH_SIZE counts 128-B requests as 64 B; '
                      'both counters in KiB; one --pmc pass each)',
           'kernels': dict(sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches_sampled']))}
    json.dump(out, open(a.out, 'w'), indent=1)
    if a.per_kernel:
        with open(a.per_kernel, 'w') as f:
            f.write('kernel,launches_sampled,fetch_size_kib_per_launch,write_size_kib_per_launch,hbm_mb_per_launch\n')
            for s, e in out['kernels'].items():
                f.write(f"{s},{e['launches_sampled']},{e['fetch_size_kib_per_launch']:.1f},{e['write_size_kib_per_launch']:.1f},"
                        f"{e['hbm_bytes_per_launch'] / 1e6:.2f}\n")
    for s, e in list(out['kernels'].items())[:14]:
        print(f"{s:34s} n {e['launches_sampled']:5d}  {e['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == '__main__':
    main()
