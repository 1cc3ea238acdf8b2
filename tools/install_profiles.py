#!/usr/bin/env python3
"""Copy what tools/refresh_profiles.sh produced (gpurun_out/prof_refresh/) into profiles/ (names truncated, noise stripped)."""
import csv, glob, shutil, collections, os
O = 'gpurun_out/prof_refresh'
shutil.copy(f'{O}/r01_bench_default.json', 'profiles/r01_bench_default.json')
shutil.copy(f'{O}/traffic.json', 'profiles/r01_pmc_traffic.json')
shutil.copy(f'{O}/r01_bench_by_kernel_and_grid.txt', 'profiles/r01_bench_by_kernel_and_grid.txt')
for f in ('r01_mesh_bench.json', 'r01_views_in_flight.txt', 'r01_uvmlp_bench.json', 'r01_volume_bench.json', 'r01_zero123_bench.json', 'r01_sds_iter_bench.json'):
    if os.path.exists(f'{O}/{f}'):
        shutil.copy(f'{O}/{f}', f'profiles/{f}')
for src, dst in ((f'{O}/r01_geometry_bench.jsonl', 'profiles/r01_geometry_bench.jsonl'), (f'{O}/r01_gemm_layers.txt', 'profiles/r01_gemm_layers.txt')):
    open(dst, 'w').writelines(l for l in open(src) if 'amdgpu.ids' not in l)
for tag, out in (('unet', 'profiles/r01_bench_kernel_stats.csv'), ('geom', 'profiles/r01_geometry_kernel_stats.csv')):
    f = glob.glob(f'{O}/{tag}/*/*kernel_stats.csv')[0]
    rows = list(csv.reader(open(f)))
    with open(out, 'w', newline='') as o:
        w = csv.writer(o)
        for r in rows:
            r[0] = r[0][:110]; w.writerow(r)
t = glob.glob(f'{O}/geom/*/*kernel_trace.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    n = r['Kernel_Name']
    if 'k_' in n and 'at::' not in n:
        agg[n[:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
with open('profiles/r01_geometry_by_kernel.txt', 'w') as f:
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v); f.write(f"{k:62s} n={len(v):4d} median {v[len(v) // 2]:9.1f} us  min {v[0]:9.1f} us\n")
print(open('profiles/r01_bench_default.json').read()[:200])
