"""Per-launch averages of the J^T J kernel from the three rocprofv3 --pmc passes made by
tools/collect_syrk_pmc.sh -> <dir>/summary.json (+ trimmed copies of the CSVs)."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
KERNEL = 'gemm_tn_f64_interior_kernel'


def load(name):
    files = glob.glob(os.path.join(d, name, '**', '*counter_collection.csv'), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


def syrk_rows(rows):
    r = [x for x in rows if KERNEL in x['Kernel_Name']]
    if not r:
        return []
    # the J^T J launch is the longest-running grid shape of this kernel
    dur = {}
    for x in r:
        dur.setdefault(int(x['Grid_Size']), []).append(int(x['End_Timestamp']) - int(x['Start_Timestamp']))
    g = max(dur, key=lambda k: sum(dur[k]) / len(dur[k]))
    return [x for x in r if int(x['Grid_Size']) == g]


out = {}
for name in ('sq', 'fetch', 'write'):
    rows = syrk_rows(load(name))
    by = {}
    disp = {}
    for x in rows:
        by.setdefault(x['Counter_Name'], {}).setdefault(x['Dispatch_Id'], 0.0)
        by[x['Counter_Name']][x['Dispatch_Id']] += float(x['Counter_Value'])
        disp[x['Dispatch_Id']] = (int(x['End_Timestamp']) - int(x['Start_Timestamp'])) / 1e6
    res = {k: sum(v.values()) / len(v) for k, v in by.items()}
    res['avg_launch_ms'] = sum(disp.values()) / max(1, len(disp))
    res['launches'] = len(disp)
    res['grid_threads'] = int(rows[0]['Grid_Size']) if rows else 0
    out[name] = res
    # trimmed CSV for profiles/
    if rows:
        with open(os.path.join(d, 'syrk_%s_pass.csv' % name), 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
sq, fe, wr = out['sq'], out['fetch'], out['write']
summary = {}
if fe.get('FETCH_SIZE'):
    summary['FETCH_SIZE_KB_raw'] = fe['FETCH_SIZE']
    summary['hbm_read_bytes_corrected'] = fe['FETCH_SIZE'] * 1024 * 2     # gfx950 x2 (MI355X_MICROARCH.md)
if wr.get('WRITE_SIZE'):
    summary['WRITE_SIZE_KB_raw'] = wr['WRITE_SIZE']
    summary['hbm_write_bytes'] = wr['WRITE_SIZE'] * 1024
if fe.get('TCC_HIT_sum') and wr.get('TCC_MISS_sum'):
    summary['tcc_hit_rate'] = fe['TCC_HIT_sum'] / (fe['TCC_HIT_sum'] + wr['TCC_MISS_sum'])
if sq.get('GRBM_GUI_ACTIVE'):
    clk = sq['GRBM_GUI_ACTIVE'] / 8.0 / (sq['avg_launch_ms'] * 1e-3)       # 8 XCDs each count
    summary['effective_clock_GHz'] = clk / 1e9
    if sq.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        # busy cycles are summed over 256 CUs x 4 SIMDs
        summary['mfma_pipe_busy_frac'] = sq['SQ_VALU_MFMA_BUSY_CYCLES'] / (sq['GRBM_GUI_ACTIVE'] / 8.0 * 256 * 4)
summary['avg_launch_ms'] = sq.get('avg_launch_ms')
summary['workgroups'] = sq.get('grid_threads', 0) // 256
out['summary'] = summary
json.dump(out, open(os.path.join(d, 'summary.json'), 'w'), indent=1)
print(json.dumps(summary, indent=1))
