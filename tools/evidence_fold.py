"""Fold the passes of tools/evidence.sh into one JSON per (tag, workload): for every kernel of the run its launches and time
per step (kernel-trace stats pass, 5 timed + 2 warm-up steps), HBM-side bytes per launch from the separate FETCH_SIZE /
WRITE_SIZE passes (KB per dispatch -> bytes; gfx950 correction FETCH_SIZE x 2 as MI355X_MICROARCH.md's HBM section
prescribes, raw values kept), the GB/s those bytes make over the stats pass's average duration against the 8 TB/s peak, and
-- when the SQ passes ran -- MFMA-pipe busy share, VALU / LDS active share, wait share and LDS bank-conflict share."""
import collections, csv, glob, hashlib, json, os, re, sys
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = os.path.join(repo, 'gpurun_out')
tag, wl = sys.argv[1], sys.argv[2]
STEPS = 7.0   # the stats pass runs 2 warm-up + 5 timed steps


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*$', '', name).strip()


stats = {}
for f in glob.glob(os.path.join(root, f'{tag}_{wl}_stats', '*', '*kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        k = short(r['Name'])
        d = stats.setdefault(k, {'calls': 0, 'total_ns': 0.0})
        d['calls'] += int(r['Calls'])
        d['total_ns'] += float(r['TotalDurationNs'])
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ['FETCH_SIZE', 'WRITE_SIZE', 'SQA', 'SQB']:
    for f in glob.glob(os.path.join(root, f'{tag}_{wl}_pmc_{c}', '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            pmc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(repo, 'eo_vae_amd', 'csrc', '*'))):
    h.update(os.path.basename(f).encode())
    h.update(open(f, 'rb').read())
out = {'command': f'tools/evidence.sh {tag} {wl}: rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ* '
                  f'(separate passes) -- python3 bench.py --workload {wl} --no-cpu-baseline',
       'units': 'FETCH_SIZE / WRITE_SIZE: KB per dispatch -> bytes; corrected = 2 x fetch + write (gfx950); GB/s over the stats '
                'pass\'s average duration; peak 8000 GB/s',
       'csrc_digest': h.hexdigest()[:16], 'conv_precision': os.environ.get('EOVAE_CONV_PRECISION', 'f16c8'), 'kernels': {}}
tot = sum(d['total_ns'] for d in stats.values())
for k, d in sorted(stats.items(), key=lambda kv: -kv[1]['total_ns']):
    if d['total_ns'] < 0.0005 * tot and k not in pmc:
        continue
    avg_ns = d['total_ns'] / max(d['calls'], 1)
    e = {'launches_per_step': round(d['calls'] / STEPS, 2), 'ms_per_step': round(d['total_ns'] / STEPS / 1e6, 4),
         'avg_us': round(avg_ns / 1e3, 2)}
    p = pmc.get(k, {})
    f_, w_ = p.get('FETCH_SIZE', []), p.get('WRITE_SIZE', [])
    if f_ and w_:
        fm, wm = 1024.0 * sum(f_) / len(f_), 1024.0 * sum(w_) / len(w_)
        e.update({'fetch_bytes_raw_per_launch': round(fm), 'write_bytes_per_launch': round(wm),
                  'traffic_bytes_per_launch_raw': round(fm + wm), 'traffic_bytes_per_launch_corrected': round(2 * fm + wm),
                  'hbm_gbps_raw': round((fm + wm) / avg_ns, 1), 'hbm_gbps_corrected': round((2 * fm + wm) / avg_ns, 1),
                  'hbm_frac_of_peak_corrected': round((2 * fm + wm) / avg_ns / 8000.0, 4)})
    if p.get('SQ_WAVE_CYCLES'):
        per = {c: sum(v) / len(v) for c, v in p.items() if c.startswith('SQ_') or c.startswith('GRBM')}
        wc = per['SQ_WAVE_CYCLES'] * 4.0
        s = {'valu_active_share': round(per.get('SQ_ACTIVE_INST_VALU', 0.0) * 4.0 / wc, 4),
             'lds_active_share': round(per.get('SQ_ACTIVE_INST_LDS', 0.0) * 4.0 / wc, 4),
             'wait_any_share': round(per.get('SQ_WAIT_ANY', 0.0) * 4.0 / wc, 4),
             'wait_inst_any_share': round(per.get('SQ_WAIT_INST_ANY', 0.0) * 4.0 / wc, 4),
             'valu_insts_per_dispatch': round(per.get('SQ_INSTS_VALU', 0.0))}
        if per.get('GRBM_GUI_ACTIVE'):
            cyc = per['GRBM_GUI_ACTIVE'] / 8.0     # summed over the 8 XCDs
            s['dispatch_cycles'] = round(cyc)
            s['mfma_pipe_busy_share'] = round(per.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc * 1024.0), 4)   # 1024 SIMDs
        if per.get('SQ_LDS_IDX_ACTIVE'):
            s['lds_bank_conflict_share_of_lds_cycles'] = round(per.get('SQ_LDS_BANK_CONFLICT', 0.0) / per['SQ_LDS_IDX_ACTIVE'], 4)
            s['lds_insts_per_dispatch'] = round(per.get('SQ_INSTS_LDS', 0.0))
            s['vmem_insts_per_dispatch'] = round(per.get('SQ_INSTS_VMEM', 0.0))
        e['sq'] = s
    out['kernels'][k] = e
out['kernel_ms_per_step_total'] = round(tot / STEPS / 1e6, 3)
if wl == 'cfg2':
    # bench.py's `roofline.traffic` reads profiles/r*_pmc_hbm_traffic.json (kernels keyed by base name, all instantiations of a
    # template together, per-launch means weighted by launches) and quotes it only when csrc_digest matches the build
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k, p_ in pmc.items():
        f_, w_ = p_.get('FETCH_SIZE', []), p_.get('WRITE_SIZE', [])
        if f_ and w_:
            a_ = agg[re.sub(r'<.*$', '', k)]
            a_[0] += 1024.0 * sum(f_); a_[1] += 1024.0 * sum(w_); a_[2] += len(f_)
    old_fmt = {'command': out['command'], 'units': out['units'], 'csrc_digest': out['csrc_digest'], 'conv_precision': out['conv_precision'],
               'kernels': {k: {'launches': n_, 'fetch_bytes_raw_per_launch': f / n_, 'write_bytes_per_launch': w / n_,
                               'traffic_bytes_per_launch_raw': (f + w) / n_, 'traffic_bytes_per_launch_corrected': (2 * f + w) / n_}
                           for k, (f, w, n_) in agg.items() if n_}}
    json.dump(old_fmt, open(os.path.join(root, f'{tag}_pmc_hbm_traffic.json'), 'w'), indent=1)
path = os.path.join(root, f'{tag}_{wl}_kernels.json')
json.dump(out, open(path, 'w'), indent=1)
print(path, len(out['kernels']), 'kernels,', out['kernel_ms_per_step_total'], 'ms of kernel time per step')
