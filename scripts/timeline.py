import csv,glob,sys
tag=sys.argv[1]
f=glob.glob("gpurun_out/prof_%s/**/*kernel_trace.csv"%tag, recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "fwgpu" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
# print the steps 20..24
n=0
start=None
for r in rows:
    nm=r["Kernel_Name"].split("(")[0].replace("void ","").replace("fwgpu::","")[:22]
    if "voice_control" in nm: n+=1
    if 60<=n<=64:
        s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
        if start is None: start=s
        print("%-22s q%s  %9.1f -> %9.1f  (%6.1f)"%(nm, r["Queue_Id"], s-start, e-start, e-s))
