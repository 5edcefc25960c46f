#!/usr/bin/env python3
"""Print the launch sequence (kernel, grid, duration) of the LAST UNet call found in a rocprofv3
kernel-trace CSV:  python tools/trace_seq.py gpurun_out/<dir>/<name>_kernel_trace.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last_in = max(i for i, n in enumerate(names) if "conv_in_kernel" in n)
end = next((i for i in range(last_in + 1, len(rows)) if "conv_out_kernel" in names[i] or "latent_step" in names[i]), len(rows) - 1)
tot = 0.0
for r in rows[last_in:end + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    short = re.sub(r"^void gc::|\(.*$", "", r["Kernel_Name"])
    print(f"{short:48s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{r['Grid_Size_Y']:>3s}x{r['Grid_Size_Z']:>3s} wg {r['Workgroup_Size_X']:>4s} lds {r.get('LDS_Block_Size', '?'):>6s} vgpr {r.get('VGPR_Count', '?'):>4s}/{r.get('Accum_VGPR_Count', '?'):>3s}  {d:8.2f} us")
span = (int(rows[end]["End_Timestamp"]) - int(rows[last_in]["Start_Timestamp"])) / 1e3
print(f"sum of kernels {tot:.1f} us, span {span:.1f} us, {end - last_in + 1} launches")
