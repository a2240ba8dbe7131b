"""Turns rocprofv3 outputs (gpurun_out/<dir>/bench_*.csv) into the small summaries committed under profiles/.

usage: python tools/summarize_profile.py <kernel-trace dir> <pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir> <out prefix> <generates in trace>
       python tools/summarize_profile.py --sq <SQ pass dir> [<SQ pass dir> ...] <out json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads
(MI355X_MICROARCH.md, HBM section), so fetched bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is taken as is.
"""
import collections
import csv
import json
import shutil
import sys


def sq_main(dirs, out_json):
    """Per-kernel, per-launch averages of the SQ counters of the given --pmc passes, and the ratios the design notes quote:
    matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;
    wave-cycle shares: waiting on anything (SQ_WAIT_ANY), waiting on an instruction's dependency counter (SQ_WAIT_INST_ANY; _LDS = on lgkmcnt for LDS),
    issuing (SQ_ACTIVE_INST_ANY), all over SQ_WAVE_CYCLES."""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for r in csv.DictReader(open(f'{d}/bench_counter_collection.csv')):
            a = acc[r['Kernel_Name']][r['Counter_Name']]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    keep = {}
    for k, cs in acc.items():
        if not any(w in k for w in ('gemm', 'attention', 'sample', 'conv', 'layernorm', 'cross_fold')):
            continue
        v = {c: t / n for c, (n, t) in cs.items()}
        v['launches'] = max(n for n, _ in cs.values())
        if v.get('SQ_BUSY_CU_CYCLES'):
            v['mfma_pipe_utilisation'] = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / v['SQ_BUSY_CU_CYCLES'] / 4.0
        if v.get('SQ_LDS_IDX_ACTIVE'):
            v['lds_conflict_share'] = v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE']
        if v.get('TCC_HIT_sum') is not None and v.get('TCC_MISS_sum') is not None and v['TCC_HIT_sum'] + v['TCC_MISS_sum'] > 0:
            # L2: hit rate over all its requests; what goes out on the fabric (TCC_EA0_RDREQ: to Infinity Cache / HBM -- the counters cannot tell the two apart) per launch
            v['l2_hit_rate'] = v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum'])
        if v.get('SQ_WAVE_CYCLES'):
            for c, name in (('SQ_WAIT_ANY', 'wait_any_share'), ('SQ_WAIT_INST_ANY', 'wait_inst_share'), ('SQ_WAIT_INST_LDS', 'wait_inst_lds_share'),
                            ('SQ_ACTIVE_INST_ANY', 'issue_share'), ('SQ_ACTIVE_INST_LDS', 'issue_lds_share'), ('SQ_ACTIVE_INST_VMEM', 'issue_vmem_share')):
                if c in v:
                    v[name] = v[c] / v['SQ_WAVE_CYCLES']
        keep[k[:110]] = v
    json.dump(keep, open(out_json, 'w'), indent=1)
    order = sorted(keep.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CU_CYCLES', kv[1].get('SQ_WAVE_CYCLES', 0)) * kv[1]['launches'])[:14]
    print('SQ counters per launch (top kernels by busy cycles):')
    for k, v in order:
        print(f"{k[:62]:62s} n {v['launches']:5d}  mfma {v.get('mfma_pipe_utilisation', 0):.3f}  lds-conflict {v.get('lds_conflict_share', 0):.3f}  "
              f"wait {v.get('wait_any_share', 0):.3f}  wait-inst {v.get('wait_inst_share', 0):.3f}  wait-lds {v.get('wait_inst_lds_share', 0):.3f}  issue {v.get('issue_share', 0):.3f}"
              + (f"  L2-hit {v['l2_hit_rate']:.3f}  EA-rd-req/launch {v.get('TCC_EA0_RDREQ_sum', 0):.3g} (DRAM-space {v.get('TCC_EA0_RDREQ_DRAM_sum', 0):.3g})" if 'l2_hit_rate' in v else ''))


def main():
    if sys.argv[1] == '--sq':
        return sq_main(sys.argv[2:-1], sys.argv[-1])
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
    lib_calls = sum(c for k, (c, _) in dur.items() if 'at::' in k or 'elementwise' in k or 'rocprim' in k or 'Cijk' in k)
    lib_ns = sum(t for k, (c, t) in dur.items() if 'at::' in k or 'elementwise' in k or 'rocprim' in k or 'Cijk' in k)
    all_calls, all_ns = sum(c for c, _ in dur.values()), sum(t for _, t in dur.values())
    print(f'kernel launches per generate: {all_calls / ngen:.0f} ({all_ns / ngen / 1e6:.2f} ms of kernel time); of them torch / library kernels '
          f'(incl. warm-up and setup of the traced process): {lib_calls / ngen:.1f} ({lib_ns / ngen / 1e6:.3f} ms)')
    top = dict(sorted(out.items(), key=lambda kv: -kv[1]['avg_launch_us'] * kv[1]['launches_per_generate'])[:14])
    json.dump(top, open(f'{prefix}_pmc_summary.json', 'w'), indent=1)
    for k, v in top.items():
        print(f"{v['avg_launch_us'] * v['launches_per_generate'] / 1e3:7.2f} ms/gen  {v['avg_launch_us']:9.1f} us  "
              f"fetch {v['hbm_fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['hbm_write_bytes_per_launch'] / 1e6:9.1f} MB  "
              f"{v['hbm_gbs']:7.1f} GB/s  {k[:60]}")


if __name__ == '__main__':
    main()
