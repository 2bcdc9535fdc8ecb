"""Reads a rocprofv3 kernel-trace CSV of `bench.py --force-reducer ...` and reports what the gradient exchange does to
the step: the collective (RCCL) kernels of the LAST step with launch-to-finish times and what ran beside them, the
share of the step in which no kernel ran, and the per-kernel totals.
usage: trace_reducer.py <kernel_trace.csv> [steps]"""
import csv
import re
import sys


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    ev = []
    for r in rows:
        name = r.get('Kernel_Name') or r.get('Name') or ''
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'^void ', '', name).replace('(anonymous namespace)::', '')[:60], r.get('Queue_Id', ''), r.get('Stream_Id', '')))
    ev.sort()
    is_coll = lambda n: 'nccl' in n.lower() or 'rccl' in n.lower() or 'oneRankReduce' in n or 'oneRank' in n
    adam = [i for i, e in enumerate(ev) if 'adamw_multi' in e[2]]
    if len(adam) < 2:
        print('no AdamW launches found: cannot cut steps')
        return
    lo, hi = adam[-2] + 1, adam[-1] + 1                        # the last full step: after the previous update up to its own
    step = ev[lo:hi]
    t0, t1 = step[0][0], max(e[1] for e in step)
    print(f'last step: {len(step)} launches, {(t1 - t0) / 1e6:.2f} ms from first kernel start to last kernel end')
    # busy union
    busy, cur0, cur1 = 0, None, None
    for s, e, *_ in step:
        if cur1 is None or s > cur1:
            if cur1 is not None:
                busy += cur1 - cur0
            cur0, cur1 = s, e
        else:
            cur1 = max(cur1, e)
    busy += cur1 - cur0
    print(f'  some kernel running: {busy / 1e6:.2f} ms; nothing running: {(t1 - t0 - busy) / 1e6:.2f} ms')
    qs = {}
    for e in step:
        qs[(e[4], e[3])] = qs.get((e[4], e[3]), 0) + 1
    print('  launches per (stream, hardware queue):', ', '.join(f'stream {k[0]} on queue {k[1]}: {v}' for k, v in sorted(qs.items())))
    coll = [e for e in step if is_coll(e[2])]
    print(f'  collective kernels: {len(coll)}')
    for s, e, n, q, st in coll:
        beside = [x for x in step if not is_coll(x[2]) and x[0] < e and x[1] > s]
        names = {}
        for x in beside:
            names[x[2][:28]] = names.get(x[2][:28], 0) + 1
        top = ', '.join(f'{k} x{v}' for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:3])
        print(f'    +{(s - t0) / 1e6:7.2f} ms  {(e - s) / 1e3:9.1f} us  {n[:40]:40s} beside: {top}')
    tot = {}
    for s, e, n, *_ in step:
        a = tot.setdefault(n, [0, 0])
        a[0] += e - s
        a[1] += 1
    print('  per kernel, last step:')
    for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:18]:
        print(f'    {d / 1e6:7.2f} ms {c:5d} x {d / c / 1e3:8.1f} us  {n}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
