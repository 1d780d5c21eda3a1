"""Fold the SQ passes of tools/pmc_sq3.sh into one table per kernel: MFMA-pipe busy share, VALU / LDS active share, wait
share, LDS bank-conflict share, instructions per wave, effective clock."""
import collections, csv, glob, json, os, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(list)
KERNELS = ['conv3x3_f16c8_w4_kernel', 'conv3x3_f16c8_kernel', 'wgrad3x3_f16c8_w4_kernel', 'wgrad3x3_f16c8_rows_kernel', 'wgrad3x3_f16c8_kernel']
for f in glob.glob(os.path.join(root, f'pmcsq3?_{tag}_*', '*', '*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = next((k for k in KERNELS if k in r['Kernel_Name']), None)
        if k is None:
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
out = {}
for k in KERNELS:
    a = acc.get(k)
    if not a:
        continue
    per = {c: a[c] / cnt[k][c] for c in a}      # per dispatch
    wc = per.get('SQ_WAVE_CYCLES', 0.0) * 4.0    # quad-cycles -> cycles, summed over waves
    busy = per.get('SQ_BUSY_CYCLES', 0.0)
    d = {'dispatches': cnt[k].get('SQ_WAVE_CYCLES', 0), 'per_dispatch': {c: round(v, 1) for c, v in per.items()}}
    if per.get('GRBM_GUI_ACTIVE'):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 = the dispatch's length in shader cycles; 1024 SIMDs on the chip
        cyc = per['GRBM_GUI_ACTIVE'] / 8.0
        d['dispatch_cycles'] = round(cyc)
        d['mfma_pipe_busy_share'] = round(per.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * 1024.0), 4)
    if wc:
        d['valu_active_share'] = round(per.get('SQ_ACTIVE_INST_VALU', 0.0) * 4.0 / wc, 4)
        d['lds_active_share'] = round(per.get('SQ_ACTIVE_INST_LDS', 0.0) * 4.0 / wc, 4)
        d['wait_any_share'] = round(per.get('SQ_WAIT_ANY', 0.0) * 4.0 / wc, 4)
        d['wait_inst_any_share'] = round(per.get('SQ_WAIT_INST_ANY', 0.0) * 4.0 / wc, 4)
    if per.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_bank_conflict_share_of_lds_cycles'] = round(per.get('SQ_LDS_BANK_CONFLICT', 0.0) / per['SQ_LDS_IDX_ACTIVE'], 4)
    out[k] = d
path = os.path.join(root, f'{tag}_pmc_sq.json')
json.dump(out, open(path, 'w'), indent=1)
print(open(path).read())
