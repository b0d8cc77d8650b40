import csv, glob, sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tail=rows[-17:]
for r in tail:
    if 'topk_cand' in r['Kernel_Name'] or 'stream_sums' in r['Kernel_Name'] or 'select' in r['Kernel_Name']:
        print(f"dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:>3} {r['Kernel_Name'][:40]}")
