"""Per-kernel-instance summary of a rocprofv3 sqlite db (rocpd): python tools/db_kernels.py <p_results.db> [min_us]
Launches of one symbol inside a step are told apart by grid size and position (as tools/summarize_profiles.py)."""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol = "display_name" if "display_name" in scols else "kernel_name"
rows = list(cur.execute(f"select s.{namecol}, d.start, d.end, d.grid_size_x * d.grid_size_y * d.grid_size_z, d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))


def short(name):
    m = re.search(r"coskad::(?:\w+::)*(\w+(?:<[^>]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name.split("(")[0][:60]


steps = max(1, sum(1 for r in rows if short(r[0]) == "k_adam_tick"))
count = defaultdict(int)
for r in rows:
    count[short(r[0])] += 1
seen = defaultdict(int)
inst = defaultdict(list)
for name, s, e, g, w in rows:
    k = short(name)
    per = count[k] // steps if count[k] % steps == 0 else 0
    slot = seen[k] % per if per else -1
    seen[k] += 1
    inst[(k, slot, g // max(1, w))].append((e - s) / 1e3)
total = sum(sum(v) for v in inst.values())
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
print(f"steps {steps}, total GPU time {total / 1e3:.2f} ms")
tot_step = 0.0
for (k, slot, wg), d in sorted(inst.items(), key=lambda kv: -sum(kv[1])):
    avg = sum(d) / len(d)
    if slot >= 0:
        tot_step += avg
    if avg >= minus:
        print(f"{k:45s} slot {slot:2d} wg {wg:5d} calls {len(d):4d} avg {avg:8.1f} min {min(d):8.1f} share {100 * sum(d) / total:5.2f}%")
print(f"sum of in-step kernel averages: {tot_step:.1f} us")
