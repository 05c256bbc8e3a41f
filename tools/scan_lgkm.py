"""Scans hipcc -S output for places where more LDS / scalar-memory operations are in flight than the 4-bit LGKM counter
(max 15) can count.  Usage: python tools/scan_lgkm.py file.s [...]; prints per kernel the worst in-flight count and where."""
import re
import sys

for path in sys.argv[1:]:
    kern, out, worst, where, wide = None, 0, {}, {}, {}
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kern, out = m.group(1), 0
            continue
        if kern is None:
            continue
        op = t.split()[0] if t else ""
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op in ("s_memtime", "s_memrealtime"):
            out += 1
            if out > worst.get(kern, 0):
                worst[kern], where[kern] = out, n
                wide[kern] = "b128" in op or "b96" in op
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                out = min(out, int(m.group(1)))
        elif op in ("s_barrier", "s_endpgm") or op.startswith("s_cbranch") or op.startswith("s_branch"):
            pass
    for k, v in sorted(worst.items(), key=lambda kv: -kv[1]):
        if v > 15:
            print(f"{path}: {k[:70]} max in flight {v} at line {where[k]}{' (wide)' if wide[k] else ''}")
