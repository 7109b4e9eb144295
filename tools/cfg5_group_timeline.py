"""kernel timelines of single group sub-steps on a ball domain from a rocprofv3 kernel trace of tools/train_cfg5.py
   (python tools/cfg5_group_timeline.py trace.csv): k_adam-delimited stretches, two generator and two discriminator ones"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
name = lambda r: re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:44]
adams = [i for i, r in enumerate(rows) if 'k_adam' in r['Kernel_Name']]
shown = {'generator': 0, 'discriminator': 0}
for a, b in zip(adams[-120:-1], adams[-119:]):
    seg = rows[a + 1:b + 1]
    if not any('k_disc_fwd' in r['Kernel_Name'] for r in seg):
        continue
    kind = 'discriminator' if any('k_disc_rec' in r['Kernel_Name'] or 'k_disc_bwd' in r['Kernel_Name'] for r in seg) else 'generator'
    if shown[kind] == 2:
        continue
    shown[kind] += 1
    t0 = int(rows[a]['End_Timestamp'])
    print('--- one group %s sub-step (%d kernels, %.1f us from the previous update to this one)' % (kind, len(seg), (int(seg[-1]['End_Timestamp']) - t0) / 1e3))
    for r in seg:
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        print('%8.1f %8.1f %7.1f  q%-2s %-44s grid=%s' % (s, e, e - s, r['Queue_Id'], name(r), r['Grid_Size_X']))
