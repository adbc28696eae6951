import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if 'fwgpu' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
i0=max(0,len(rows)-40)
prev=None
for r in rows[i0:i0+30]:
    s,e=int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0
    gap = (s-prev)/1e3 if prev is not None else 0
    print("%-26s start %9.1f dur %7.1f gap_before %6.1f" % (r['Kernel_Name'].split('(')[0].replace('void fwgpu::','').replace('fwgpu::','')[:26], s/1e3, (e-s)/1e3, gap))
    prev=e
