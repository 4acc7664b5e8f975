import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find the LAST k_exl_factor launch and print everything from there
idx = [i for i, r in enumerate(rows) if 'k_exl_factor' in r['Kernel_Name']]
i0 = idx[-1]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%9.1f us  dur %9.1f  gap %7.1f  %s grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:60], r.get('Grid_Size', '')))
    prev_end = e
