"""print the kernel timeline of one g,g,d cycle from a rocprofv3 kernel trace:  python tools/timeline.py <kernel_trace.csv> [cycle index from the end]"""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(r'\bk_\w+', r['Kernel_Name'])]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 8
idx = [i for i, r in enumerate(rows) if 'k_disc_bwd' in r['Kernel_Name'] or 'k_disc_rec' in r['Kernel_Name']]
i0 = idx[-back - 1]
i1 = idx[-back]
t0 = int(rows[i0]['End_Timestamp'])
for r in rows[i0:i1 + 3]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    name = re.search(r'(k_\w+)(<[^>]*>)?', r['Kernel_Name']).group(0)
    print('%8.1f %8.1f %7.1f  q%-2s %-34s grid=%s' % (s, e, e - s, r['Queue_Id'], name[:34], r['Grid_Size_X']))
