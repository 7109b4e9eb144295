"""profiles/r04_shard_sweep.md's first table and profiles/r04_other_configs_1gpu.md's tables from the kept bench lines
(profiles/r04_lines/) and the HBM counters of configs[3]:  python tools/r04_tables.py"""
import json


def J(f):
    return json.loads(open(f).read().strip().splitlines()[-1])


rows = [('configs[2] WHOLE: d=50, 16384 paths, N_t=64', 'profiles/r04_lines/cfg2_d50_16384x64_1gpu.json'),
        ('configs[3] WHOLE: d=100, 65536 paths, N_t=32', 'profiles/r04_lines/cfg3_d100_65536x32_1gpu.json'),
        ('configs[2] 1/8 share: d=50, 2048 paths, N_t=64', 'profiles/r04_lines/cfg2_share_2048x64_1gpu.json'),
        ('configs[3] 1/8 share: d=100, 8192 paths, N_t=32', 'profiles/r04_lines/cfg3_share_8192x32_1gpu.json')]
print('| workload | sub-steps/s | ms / sub-step | whole-step FP64 fraction | `k_disc_fwd` as launched / solo |\n|---|---|---|---|---|')
for name, f in rows:
    o = J(f)
    print('| %s | %.1f | %.3f | %.3f | %.3f / %.3f |' % (name, o['value'], o['ms_per_step'], o['whole_step']['frac_fp64_matrix_peak'],
                                                       o['roofline']['frac'], o['roofline'].get('solo_full_grid', {}).get('frac', float('nan'))))
p = json.load(open('profiles/r04_pmc_traffic_cfg3.json'))['kernels']
per = {'k_disc_fwd<50,false,true,13>': 2, 'k_disc_fwd<50,true,true,13>': 1, 'k_disc_rec<50,0,3>': 1, 'k_ode_fwd<20,10,8,1,1>': 2,
       'k_ode_fwd<20,10,8,1,2>': 1, 'k_ode_bwd_duo<20,10,8,1>': 4, 'k_ode_bwd<20,10,8,1,false,true,false>': 1, 'k_weak_partials': 3,
       'k_adam': 3, 'k_bdry': 2, 'k_disc_cot': 1}
print('\n| kernel | HBM GB per launch | launches per g,g,d cycle |\n|---|---|---|')
tot = 0.0
for k, n in per.items():
    gb = p[k]['hbm_bytes_per_launch_corrected'] / 1e9
    tot += gb * n
    print('| `%s` | %.2f | %d |' % (k, gb, n))
o = J('profiles/r04_lines/cfg3_d100_65536x32_1gpu.json')
cyc = 3 * o['ms_per_step']
print('\nper cycle %.1f GB in %.1f ms = %.2f TB/s = %.0f %% of 8 TB/s (%.0f %% of 6.3); sub-step %.2f ms, whole-step fraction %.3f'
      % (tot, cyc, tot / cyc, 100 * tot / cyc / 8.0, 100 * tot / cyc / 6.3, o['ms_per_step'], o['whole_step']['frac_fp64_matrix_peak']))
print('\n| paths on the GPU | 16-path tiles only: sub-steps/s (ms) | default policy: sub-steps/s (ms) | gain |\n|---|---|---|---|')
for n in (4096, 2048, 1024, 512):
    a, b = J('profiles/r04_lines/d20_gp%d_narrow0.json' % n), J('profiles/r04_lines/d20_gp%d_narrow1.json' % n)
    print('| %d | %.0f (%.4f) | %.0f (%.4f) | %+.1f %% |' % (n, a['value'], a['ms_per_step'], b['value'], b['ms_per_step'], 100 * (b['value'] / a['value'] - 1)))
