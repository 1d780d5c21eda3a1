"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into one JSON (HBM bytes per launch of the
two dominant kernels; gfx950 correction FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes, raw values kept)."""
import collections, csv, glob, hashlib, json, os, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ['FETCH_SIZE', 'WRITE_SIZE']:
    for f in glob.glob(os.path.join(root, f'{tag}_pmc_{c}', '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            for k in ['conv3x3_f16c8_q16_kernel', 'conv3x3_f16c8_w4_kernel', 'conv3x3_f16c8_kernel', 'conv3x3_bf16_kernel', 'wgrad3x3_bf16_kernel', 'wgrad3x3_f16c8_rows_kernel', 'wgrad3x3_f16c8_kernel', 'gemm_bf16_kernel',
                      'gn_apply_packed_kernel', 'gn_bwd_fused_pipe_kernel<true, true', 'gn_bwd_fused_pipe_kernel<true, false',
                      'gn_bwd_fused_kernel', 'gn_bwd_reduce_kernel', 'gn_bwd_apply_kernel']:
                if k in r['Kernel_Name']:
                    acc[k][c].append(float(r['Counter_Value']) * 1024.0)
out = {'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline',
       'units': 'FETCH_SIZE / WRITE_SIZE are KB per dispatch; bytes = KB*1024; gfx950 correction: FETCH_SIZE x2 (MI355X_MICROARCH.md HBM section)',
       'kernels': {}}
repo = os.path.dirname(root)
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(repo, 'eo_vae_amd', 'csrc', '*'))):   # the build these numbers belong to (bench.py checks it)
    h.update(os.path.basename(f).encode())
    h.update(open(f, 'rb').read())
out['csrc_digest'] = h.hexdigest()[:16]
out['conv_precision'] = os.environ.get('EOVAE_CONV_PRECISION', 'f16c8')   # arithmetic mode of the profiled run
for k, d in acc.items():
    f, w = d.get('FETCH_SIZE', []), d.get('WRITE_SIZE', [])
    if not f or not w:
        continue
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    out['kernels'][k] = {'launches': len(f), 'fetch_bytes_raw_per_launch': fm, 'write_bytes_per_launch': wm,
                         'fetch_bytes_raw_sum': sum(f), 'write_bytes_sum': sum(w),
                         'traffic_bytes_per_launch_raw': fm + wm, 'traffic_bytes_per_launch_corrected': 2 * fm + wm}
path = os.path.join(root, f'{tag}_pmc_hbm_traffic.json')
json.dump(out, open(path, 'w'), indent=1)
print(open(path).read())
