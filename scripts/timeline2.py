"""kernel timeline of steps a..b of a rocprofv3 --kernel-trace run: python scripts/timeline2.py <tag> [first_step] [n_steps]
(a step starts at its first control / leaf kernel); prints start, end, duration and the gap to the kernel before"""
import csv, glob, sys
tag = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob("gpurun_out/prof_%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fwgpu" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n, prev_end, start = 0, None, None
for r in rows:
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fwgpu::", "")[:26]
    if "k_root_out" in nm or "k_graph_out" in nm:
        pass
    s, e = int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3
    if first <= n < first + nst:
        if start is None:
            start = s
        print("%-26s q%-3s %9.1f -> %9.1f  (%7.1f)  gap %6.1f" % (nm, r["Queue_Id"], s - start, e - start, e - s, (s - prev_end) if prev_end else 0.0))
    prev_end = max(prev_end or 0, e)
    if "k_root_out" in nm or "k_graph_out" in nm:
        n += 1
