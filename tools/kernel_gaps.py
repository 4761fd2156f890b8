"""The last kernels of a rocprofv3 --kernel-trace run as a timeline: start (us), duration, gap to the previous kernel's end, queue, grid, name.
    python3 tools/kernel_gaps.py <rocprofv3 output dir> [kernels = 24]"""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
tail = rows[-n-40:-40]
t0 = int(tail[0]['Start_Timestamp'])
prev_end = t0
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.2f} {(e-s)/1e3:8.2f} gap {(s-prev_end)/1e3:7.2f} q{r.get('Queue_Id','?')} grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size','?'))} vgpr {r.get('VGPR_Count', '?')} lds {r.get('LDS_Block_Size','?')} {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end, e)
