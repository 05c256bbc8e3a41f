"""Prints the last N launches of a rocprofv3 kernel trace (directory given): dur / gap / workgroups / kernel -- e.g. the eval forwards a
leg of bench.py ends with.  usage: python tools/tail_timeline.py <trace dir> [N]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(name):
    m = re.search(r"coskad::(?:\w+::)*(\w+(?:<[^>]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name.split("(")[0][:70]
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"dur {(e - s) / 1e3:7.1f} gap {((s - prev) / 1e3 if prev else 0):7.1f} wg {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):6d} {short(r['Kernel_Name'])}")
    prev = e
