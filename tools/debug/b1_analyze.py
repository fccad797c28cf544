"""Developer tool: per-kernel shares of the LAST DDIM-50 run in a rocprofv3 kernel trace of tools/debug/b1_trace.py."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/b1trace/b1_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
fills = [i for i, r in enumerate(rows) if 'ddim_step_kernel' in r['Kernel_Name']]
# 50 steps per run, last run = last 50 ddim steps; start at the temb kernel preceding the first of them
first = fills[-50]
start = max(i for i in range(first) if 'temb_kernel' in rows[i]['Kernel_Name'])
seg = rows[start:fills[-1] + 2]
def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('void ', '')
    depth = 0; out = []
    for ch in n:
        if ch == '<': depth += 1
        if ch == '(' and depth == 0: break
        if ch == '>': depth -= 1
        out.append(ch)
    return ''.join(out)[:100]
c = collections.Counter(); d = collections.defaultdict(float)
for r in seg:
    k = short(r['Kernel_Name']); c[k] += 1; d[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(d.values())
print("launches", len(seg), "busy ms", round(tot / 1e3, 2), "wall ms", round((int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6, 2))
for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:24]:
    print(f"{100*v/tot:5.1f}% {c[k]:6d} {c[k]/50:6.1f}/step {v/c[k]:8.2f} us  {k}")
if len(sys.argv) > 2:
    idx = [i for i, r in enumerate(seg) if 'temb_kernel' in r['Kernel_Name']]
    for r in seg[idx[20]:idx[21]]:
        print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:7.2f} us grid ({int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])},{r['Grid_Size_Y']}) vgpr {r['VGPR_Count']:>4} {short(r['Kernel_Name'])[:70]}")
