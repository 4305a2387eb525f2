import csv, glob, sys
for tag in sys.argv[1:]:
    for f in glob.glob(f"/tmp/kp_{tag}/*/*_kernel_trace.csv"):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"][:110], row.get("LDS_Block_Size"), row.get("Scratch_Size"), row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("Workgroup_Size"), row.get("Grid_Size"))
            if k not in seen:
                seen.add(k); print(tag, k)
