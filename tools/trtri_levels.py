"""Per-launch durations of the L^-1 kernels (and k_grad) in the LAST evaluation of a rocprofv3 kernel trace (csv) of tools/eval_once.py:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/x -- python3 tools/eval_once.py 8192 5
    python tools/trtri_levels.py gpurun_out/x/.../*_kernel_trace.csv
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = [i for i, r in enumerate(rows) if 'k_gram' in r['Kernel_Name']][-1]
t0 = int(rows[first]['Start_Timestamp'])
total = 0.0
for r in rows[first:]:
    n = r['Kernel_Name']
    if 'trtri' in n or 'inv128' in n or 'k_grad' in n:
        s = (int(r['Start_Timestamp']) - t0) / 1e3
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if 'k_grad' not in n:
            total += d
        print(f"{s:9.1f} {d:8.1f} {n.split('(')[0][:32]:32s} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
print(f'L^-1 kernels summed: {total:.1f} us')
