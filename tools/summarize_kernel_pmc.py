"""Per-kernel, per-launch averages of the rocprofv3 --pmc passes made by tools/collect_kernel_pmc.sh:
<dir>/<tag>/<pass>/**/counter_collection.csv -> <dir>/<tag>_pmc.json.  For every kernel that takes more than 0.5 % of its
program's kernel time: launches, average duration, the counters, and the split of its wave-cycles into
issue (SQ_ACTIVE_INST_ANY) / dependency stall (SQ_WAIT_INST_ANY, of which SQ_WAIT_INST_LDS) / parked (SQ_WAIT_ANY: s_waitcnt,
barriers), MFMA pipe busy share, effective clock, fabric bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE) and L2 hit rate."""
import csv
import glob
import json
import os
import re
import sys

d = sys.argv[1]
N_CU, N_SIMD = 256, 4


def short(name):
    name = re.sub(r'\(.*$', '', name)            # argument list
    name = re.sub(r'^void\s+', '', name)
    name = name.replace('lsqamd::', '').replace('(anonymous namespace)::', '')
    return name.strip()


def load(tag, p):
    rows = []
    for f in glob.glob(os.path.join(d, tag, p, '**', '*counter_collection.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel(rows):
    """kernel -> {'launches', 'avg_ms', counter: per-launch average}"""
    acc = {}
    for x in rows:
        k = short(x['Kernel_Name'])
        e = acc.setdefault(k, {'disp': {}, 'ctr': {}})
        e['disp'][x['Dispatch_Id']] = (int(x['End_Timestamp']) - int(x['Start_Timestamp'])) / 1e6
        e['ctr'].setdefault(x['Counter_Name'], {}).setdefault(x['Dispatch_Id'], 0.0)
        e['ctr'][x['Counter_Name']][x['Dispatch_Id']] += float(x['Counter_Value'])
    out = {}
    for k, e in acc.items():
        n = len(e['disp'])
        r = {'launches': n, 'avg_ms': sum(e['disp'].values()) / n, 'total_ms': sum(e['disp'].values())}
        for c, v in e['ctr'].items():
            r[c] = sum(v.values()) / len(v)
        out[k] = r
    return out


for tag in sorted(os.listdir(d)):
    if not os.path.isdir(os.path.join(d, tag)):
        continue
    passes = {p: per_kernel(load(tag, p)) for p in ('sq', 'sq2', 'fetch', 'write')}
    sq = passes['sq']
    if not sq:
        continue
    total = sum(v['total_ms'] for v in sq.values())
    res = {}
    for k, v in sorted(sq.items(), key=lambda kv: -kv[1]['total_ms']):
        if v['total_ms'] < 0.005 * total:
            continue
        e = {'launches': v['launches'], 'avg_us': v['avg_ms'] * 1e3, 'share_of_kernel_time': v['total_ms'] / total}
        wc = v.get('SQ_WAVE_CYCLES', 0.0)
        if wc > 0:
            e['wave_cycles_split'] = {
                'issue': v.get('SQ_ACTIVE_INST_ANY', 0.0) / wc,
                'dependency_stall': v.get('SQ_WAIT_INST_ANY', 0.0) / wc,
                'of_which_lds_issue_stall': v.get('SQ_WAIT_INST_LDS', 0.0) / wc,
                'parked_waitcnt_barrier': v.get('SQ_WAIT_ANY', 0.0) / wc,
            }
        if v.get('GRBM_GUI_ACTIVE'):
            clk = v['GRBM_GUI_ACTIVE'] / 8.0                                  # 8 XCDs each count
            e['effective_clock_GHz'] = clk / (v['avg_ms'] * 1e-3) / 1e9
            e['mfma_pipe_busy_frac_of_chip'] = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (clk * N_CU * N_SIMD)
            # occupancy-weighted: wave-cycles are quad-cycles summed over waves
            e['avg_waves_in_flight'] = 4.0 * wc / clk if clk else None
        e['lds_bank_conflict_cycles'] = v.get('SQ_LDS_BANK_CONFLICT')
        s2 = passes['sq2'].get(k)
        if s2:
            for c in ('SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_ACTIVE_INST_VALU',
                      'SQ_ACTIVE_INST_LDS', 'SQ_BUSY_CYCLES'):
                if c in s2:
                    e[c] = s2[c]
        fe, wr = passes['fetch'].get(k), passes['write'].get(k)
        if fe and fe.get('FETCH_SIZE') is not None:
            e['fabric_read_bytes'] = fe['FETCH_SIZE'] * 1024 * 2              # gfx950 x2 (MI355X_MICROARCH.md)
        if wr and wr.get('WRITE_SIZE') is not None:
            e['fabric_write_bytes'] = wr['WRITE_SIZE'] * 1024
        if fe and wr and fe.get('TCC_HIT_sum') and wr.get('TCC_MISS_sum') is not None:
            e['l2_hit_rate'] = fe['TCC_HIT_sum'] / (fe['TCC_HIT_sum'] + wr['TCC_MISS_sum'])
        if 'fabric_read_bytes' in e:
            e['fabric_TBps'] = (e['fabric_read_bytes'] + e.get('fabric_write_bytes', 0.0)) / (v['avg_ms'] * 1e-3) / 1e12
        res[k] = e
    json.dump(res, open(os.path.join(d, tag + '_pmc.json'), 'w'), indent=1)
    print('==', tag)
    for k, e in res.items():
        s = e.get('wave_cycles_split', {})
        print('%-70s n=%4d %9.1f us  issue %.2f dep %.2f (lds %.2f) parked %.2f  mfma %.3f  waves %.0f' % (
            k[:70], e['launches'], e['avg_us'], s.get('issue', 0), s.get('dependency_stall', 0), s.get('of_which_lds_issue_stall', 0),
            s.get('parked_waitcnt_barrier', 0), e.get('mfma_pipe_busy_frac_of_chip', 0) or 0, e.get('avg_waves_in_flight', 0) or 0))
