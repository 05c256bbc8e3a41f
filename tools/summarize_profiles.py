"""Turns the rocprofv3 output of tools/collect_profiles.sh (under gpurun_out/<tag>/) into the committed summaries:
  profiles/<tag>_bench_kernel_stats.csv   rocprofv3's own --stats table (per kernel symbol)
  profiles/<tag>_kernel_instances.csv     per (kernel, k-th launch of that kernel inside a step, grid) = per layer: calls, avg/min us, share of GPU time
  profiles/<tag>_hbm_traffic_pmc.csv      FETCH_SIZE / WRITE_SIZE per instance (KB, averaged over dispatches)
  profiles/<tag>_hbm_traffic.json         HBM bytes / launch for the kernels bench.py prices: (2*FETCH_SIZE + WRITE_SIZE) * 1024
                                          (gfx950: FETCH_SIZE reports half of a wide coalesced stream; guide's correction)
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    m = re.search(r"coskad::(?:\w+::)*(\w+(?:<[^>]*>)?)", name)
    return m.group(1).replace(" ", "") if m else name.split("(")[0]


def one(pattern: str) -> str:
    hits = glob.glob(pattern, recursive=True)
    if not hits:
        sys.exit(f"missing {pattern}")
    return hits[0]


def main(tag: str) -> None:
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(one(f"{src}/stats/**/*kernel_stats.csv"), f"{dst}/{tag}_bench_kernel_stats.csv")

    # dynamic LDS is not in the trace, so launches of one symbol are told apart by their position inside a step:
    # the k-th launch of a symbol in every step is the same layer ("slot" k; steps = number of k_adam_tick launches)
    def slotted(rows, name_col):
        rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
        steps = max(1, sum(1 for r in rows if short(r[name_col]) == "k_adam_tick"))
        count = defaultdict(int)
        for r in rows:
            count[short(r[name_col])] += 1
        seen = defaultdict(int)
        for r in rows:
            k = short(r[name_col])
            per = count[k] // steps if count[k] % steps == 0 else 0
            slot = seen[k] % per if per else -1
            seen[k] += 1
            yield k, slot, r

    inst = defaultdict(list)
    with open(one(f"{src}/stats/**/*kernel_trace.csv")) as f:
        for k, slot, r in slotted(list(csv.DictReader(f)), "Kernel_Name"):
            key = (k, slot, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
            inst[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in inst.values())
    rows = sorted(inst.items(), key=lambda kv: -sum(kv[1]))
    with open(f"{dst}/{tag}_kernel_instances.csv", "w") as f:
        f.write("kernel,launch_slot_in_step,workgroups,calls,avg_us,min_us,share_pct\n")
        for (k, lds, wg), d in rows:
            f.write(f"\"{k}\",{lds},{wg},{len(d)},{sum(d) / len(d):.1f},{min(d):.1f},{100 * sum(d) / total:.2f}\n")

    pmc = defaultdict(lambda: defaultdict(list))
    for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        with open(one(f"{src}/{sub}/**/*counter_collection.csv")) as f:
            for k, slot, r in slotted([r for r in csv.DictReader(f) if r["Counter_Name"] == counter], "Kernel_Name"):
                key = (k, slot, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
                pmc[key][counter].append(float(r["Counter_Value"]))
    summary = {}
    with open(f"{dst}/{tag}_hbm_traffic_pmc.csv", "w") as f:
        f.write("kernel,launch_slot_in_step,workgroups,dispatches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_MB_per_launch\n")
        for key, c in sorted(pmc.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
            fe = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"]))
            wr = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"]))
            hbm = (2 * fe + wr) * 1024
            summary[key] = (fe, wr, hbm)
            f.write(f"\"{key[0]}\",{key[1]},{key[2]},{len(c['FETCH_SIZE'])},{fe:.1f},{wr:.1f},{hbm / 1e6:.1f}\n")

    # matrix-pipe utilisation and LDS bank conflicts per kernel instance (optional passes)
    extra = defaultdict(lambda: defaultdict(list))
    for sub in ("mfma", "lds"):
        hits = glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True)
        if not hits:
            continue
        with open(hits[0]) as f:
            rows_ = list(csv.DictReader(f))
        names = sorted({r["Counter_Name"] for r in rows_})
        for cname in names:
            for k, slot, r in slotted([r for r in rows_ if r["Counter_Name"] == cname], "Kernel_Name"):
                extra[(k, slot, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))][cname].append(float(r["Counter_Value"]))
    if extra:
        with open(f"{dst}/{tag}_mfma_lds_pmc.csv", "w") as f:
            f.write("kernel,launch_slot_in_step,workgroups,MfmaUtil_pct,lds_bank_conflict_pct_of_lds_active\n")
            def avg(c, n):
                return sum(c.get(n, [0.0])) / max(1, len(c.get(n, [])))
            order = sorted(extra.items(), key=lambda kv: -avg(kv[1], "MfmaUtil"))
            for key, c in order:
                if not key[0].startswith("k_"):
                    continue
                conf, idx = avg(c, "SQ_LDS_BANK_CONFLICT"), avg(c, "SQ_LDS_IDX_ACTIVE")
                util = avg(c, "MfmaUtil") if "MfmaUtil" in c else float("nan")   # rocprofv3's derived metric, per launch
                lds = 100.0 * conf / idx if idx else float("nan")
                f.write(f"\"{key[0]}\",{key[1]},{key[2]},{util:.1f},{lds:.1f}\n")

    # wave-cycle breakdown (optional pass): waits / issue stalls / active, as a share of SQ_WAVE_CYCLES per kernel symbol
    hits = glob.glob(f"{src}/sq/**/*counter_collection.csv", recursive=True)
    if hits:
        sq = defaultdict(lambda: defaultdict(list))
        with open(hits[0]) as f:
            for r in csv.DictReader(f):
                sq[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        names = ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU")
        rows_ = []
        for k, c in sq.items():
            wc = sum(c.get("SQ_WAVE_CYCLES", [0.0]))
            if wc > 0 and k.startswith("k_"):
                rows_.append((wc, k, [100.0 * sum(c.get(n, [0.0])) / wc for n in names]))
        with open(f"{dst}/{tag}_sq_wait_pmc.csv", "w") as f:
            f.write("kernel,wait_any_pct,wait_inst_any_pct,wait_inst_lds_pct,active_inst_any_pct,active_inst_lds_pct,active_inst_valu_pct\n")
            for _, k, v in sorted(rows_, reverse=True):
                f.write(f"\"{k}\"," + ",".join(f"{x:.1f}" for x in v) + "\n")

    def biggest(prefix: str):
        c = [(k, v) for k, v in summary.items() if k[0].startswith(prefix)]
        return max(c, key=lambda kv: kv[1][2]) if c else None

    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/collect_profiles.sh, B=4096); "
                   "counters are KB; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE reads "
                   "1/2 of a wide coalesced stream; narrower accesses uncalibrated: upper bound)"}
    for label, prefix in (("bwd_data layer4", "k_bwd_data_f<12,17,2,1>"), ("bwd_fused layer4", "k_layer_bwd_bpc<2,4"),
                          ("bwd_fused layer3", "k_layer_bwd_bpc<1,2"), ("bwd_fused layer2", "k_layer_bwd_bpc<2,1"),
                          ("layer_apply layer4", "k_layer_apply_bpc<2>"), ("fused_encoder", "k_fused_encoder"),
                          ("bwd_stats layer4", "k_bwd_stats_bpc<2,4>"), ("apply_next layer1", "k_layer_apply_next_bpc<0,2>"),
                          ("apply_next layer2", "k_layer_apply_next_bpc<2,1>"), ("apply_next layer3", "k_layer_apply_next_bpc<1,2>")):
        b = biggest(prefix)
        if b:
            (k, lds, wg), (fe, wr, hbm) = b
            out[label] = {"kernel": k, "launch_slot_in_step": lds, "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
                          "hbm_bytes_per_launch": int(hbm)}
    if "bwd_fused layer4" in out and "bwd_stats layer4" in out:
        # bench.py's layer-level roofline: the two kernels that move layer 4's backward bytes (folds / partial-row sums: < 1 MB)
        out["layer4 backward"] = {"kernels": [out["bwd_stats layer4"]["kernel"], out["bwd_fused layer4"]["kernel"]],
                                  "hbm_bytes_per_launch": out["bwd_stats layer4"]["hbm_bytes_per_launch"] + out["bwd_fused layer4"]["hbm_bytes_per_launch"]}
    with open(f"{dst}/{tag}_hbm_traffic.json", "w") as f:
        json.dump(out, f, indent=1)
    print(open(f"{dst}/{tag}_kernel_instances.csv").read())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
