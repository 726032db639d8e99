"""Print the kernel timeline of the last batch of each rocprofv3 kernel trace under gpurun_out/prof_r2/<name>."""
import csv, glob, sys
for name in sys.argv[1:]:
    # (a name under gpurun_out/prof_r2, or any directory that holds a rocprofv3 kernel trace)
    f = (glob.glob(f'{name}/**/*kernel_trace.csv', recursive=True) if '/' in name else glob.glob(f'gpurun_out/prof_r2/{name}/*kernel_trace.csv'))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if "k_count" in r["Kernel_Name"] or "k_plan" in r["Kernel_Name"]]
    i0, i1 = idx[-2], idx[-1]
    t0 = int(rows[i0]['Start_Timestamp'])
    print('==', name)
    for r in rows[i0:i1]:
        s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
        print(f"  {r['Kernel_Name'][:34]:34s} start {s/1e3:8.1f} end {e/1e3:8.1f} dur {(e-s)/1e3:7.1f} us")
    print('  batch period', (int(rows[i1]['Start_Timestamp']) - t0) / 1e3, 'us')
