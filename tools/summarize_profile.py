"""Turns rocprofv3 outputs (gpurun_out/<dir>/bench_*.csv) into the small summaries committed under profiles/.

usage: python tools/summarize_profile.py <kernel-trace dir> <pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir> <out prefix> <generates in trace>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so fetched bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as is.
"""
import collections
import csv
import json
import shutil
import sys


def main():
    trace_dir, fetch_dir, write_dir, prefix, ngen = sys.argv[1:6]
    ngen = int(ngen)
    shutil.copy(f'{trace_dir}/bench_kernel_stats.csv', f'{prefix}_kernel_stats.csv')
    pmc = collections.defaultdict(lambda: dict(launches=0, fetch_kib=0.0, write_kib=0.0))
    for d, key in ((fetch_dir, 'fetch_kib'), (write_dir, 'write_kib')):
        for r in csv.DictReader(open(f'{d}/bench_counter_collection.csv')):
            k = r['Kernel_Name']
            pmc[k][key] += float(r['Counter_Value'])
            if key == 'fetch_kib':
                pmc[k]['launches'] += 1
    dur = {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f'{trace_dir}/bench_kernel_stats.csv'))}
    out = {}
    for k, v in pmc.items():
        if v['launches'] == 0 or k not in dur:
            continue
        calls, tot = dur[k]
        fetch_b = 2.0 * v['fetch_kib'] * 1024 / v['launches']
        write_b = v['write_kib'] * 1024 / v['launches']
        out[k] = dict(launches_per_generate=calls / ngen, avg_launch_us=tot / calls / 1e3,
                      hbm_fetch_bytes_per_launch=fetch_b, hbm_write_bytes_per_launch=write_b,
                      hbm_bytes_per_launch=fetch_b + write_b,
                      hbm_gbs=(fetch_b + write_b) / (tot / calls))
    top = dict(sorted(out.items(), key=lambda kv: -kv[1]['avg_launch_us'] * kv[1]['launches_per_generate'])[:14])
    json.dump(top, open(f'{prefix}_pmc_summary.json', 'w'), indent=1)
    for k, v in top.items():
        print(f"{v['avg_launch_us'] * v['launches_per_generate'] / 1e3:7.2f} ms/gen  {v['avg_launch_us']:9.1f} us  "
              f"fetch {v['hbm_fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['hbm_write_bytes_per_launch'] / 1e6:9.1f} MB  "
              f"{v['hbm_gbs']:7.1f} GB/s  {k[:60]}")


if __name__ == '__main__':
    main()
