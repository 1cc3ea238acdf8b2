#!/usr/bin/env python3
"""Copy what tools/refresh_profiles.sh produced (gpurun_out/prof_refresh/) into profiles/ (names truncated, noise stripped)."""
import csv, glob, shutil, collections, os
O = 'gpurun_out/prof_refresh'
R = os.environ.get('ROUND', 'r03')
shutil.copy(f'{O}/{R}_bench_default.json', f'profiles/{R}_bench_default.json')
shutil.copy(f'{O}/traffic.json', f'profiles/{R}_pmc_traffic.json')
shutil.copy(f'{O}/{R}_bench_by_kernel_and_grid.txt', f'profiles/{R}_bench_by_kernel_and_grid.txt')
for f in (f'{R}_mesh_bench.json', f'{R}_views_in_flight.txt', f'{R}_uvmlp_bench.json', f'{R}_volume_bench.json', f'{R}_zero123_bench.json', f'{R}_sds_loop_bench.json',
          f'{R}_mesh_batch_bench.json', f'{R}_bench_mesh_mode.json', f'{R}_gemm_square.txt', f'{R}_views_batched.txt', f'{R}_uvmlp_bench_exact_f32.json'):
    if os.path.exists(f'{O}/{f}'):
        shutil.copy(f'{O}/{f}', f'profiles/{f}')
for src, dst in ((f'{O}/{R}_geometry_bench.jsonl', f'profiles/{R}_geometry_bench.jsonl'), (f'{O}/{R}_gemm_layers.txt', f'profiles/{R}_gemm_layers.txt')):
    open(dst, 'w').writelines(l for l in open(src) if 'amdgpu.ids' not in l)
for tag, out in (('unet', f'profiles/{R}_bench_kernel_stats.csv'), ('geom', f'profiles/{R}_geometry_kernel_stats.csv')):
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
with open(f'profiles/{R}_geometry_by_kernel.txt', 'w') as f:
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v); f.write(f"{k:62s} n={len(v):4d} median {v[len(v) // 2]:9.1f} us  min {v[0]:9.1f} us\n")
if os.path.exists(f'{O}/mfma.txt'):
    shutil.copy(f'{O}/mfma.txt', f'profiles/{R}_pmc_mfma.txt')
print(open(f'profiles/{R}_bench_default.json').read()[:200])
