"""Summary of a rocprofv3 kernel + memory-copy trace of `bench.py --train`: one steady-state training step (between two launches of embed_kernel) --
the span of the step, per-stream busy time in its forward / backward / tail phases, the idle stretches of the device, the top kernels per stream.
usage: python tools/train_trace_summary.py <kernel_trace.csv> [<memory_copy_trace.csv>]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r.get('Stream_Id', ''), r['Kernel_Name']) for r in rows]
    if len(sys.argv) > 2:
        try:
            for r in csv.DictReader(open(sys.argv[2])):
                ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r.get('Stream_Id', ''), 'memcpy ' + r.get('Direction', '')))
        except OSError:
            pass
    ev.sort()
    idx = [i for i, e in enumerate(ev) if 'embed_kernel' in e[3]]
    a, b = idx[-2], idx[-1]
    step = ev[a:b]
    T0 = step[0][0]
    span = (ev[b][0] - T0) / 1e6
    ce = next(s for s, d, st, n in step if 'ce_bwd' in n)
    eb = next(s + d for s, d, st, n in step if 'embed_token_bwd' in n or 'embed_bwd_token' in n)
    print(f'one training step (embed_kernel to embed_kernel): {span:.3f} ms, {len(step)} launches / copies')
    print(f'  forward 0 - {(ce - T0) / 1e6:.3f} ms, backward - {(eb - T0) / 1e6:.3f} ms, tail (gradient scale, AdamW, next step\'s masking + host + parameter preparation) - {span:.3f} ms')
    iv = sorted((s, s + d) for s, d, _, _ in step)
    busy, cs, cend = 0, iv[0][0], iv[0][1]
    idle = []
    for s, e in iv[1:]:
        if s > cend:
            busy += cend - cs
            if s - cend > 20000:
                idle.append(((cend - T0) / 1e6, (s - cend) / 1e3))
            cs, cend = s, e
        else:
            cend = max(cend, e)
    busy += cend - cs
    print(f'  device busy (union over streams) {busy / 1e6:.3f} ms;  idle stretches > 20 us: ' + ', '.join(f'{d:.0f} us at {t:.2f} ms' for t, d in idle))
    for lo, hi, name in ((T0, ce, 'forward'), (ce, eb, 'backward'), (eb, ev[b][0], 'tail')):
        for st in sorted({e[2] for e in step}):
            agg = defaultdict(lambda: [0, 0])
            tot = 0
            for s, d, s_, n in step:
                if lo <= s < hi and s_ == st:
                    k = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
                    agg[k][0] += d
                    agg[k][1] += 1
                    tot += d
            if not tot:
                continue
            print(f'  {name}, stream {st}: busy {tot / 1e6:.3f} ms of {(hi - lo) / 1e6:.3f}')
            for k, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:9]:
                print(f'      {d / 1e6:7.3f} ms x{c:4d}  {k}')


if __name__ == '__main__':
    main()
