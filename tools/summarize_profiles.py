"""Turns the rocprofv3 output of tools/collect_profiles.sh (under gpurun_out/<tag>/) into the committed summaries:
  profiles/<tag>_bench_kernel_stats.csv   rocprofv3's own --stats table (per kernel symbol)
  profiles/<tag>_kernel_instances.csv     per (kernel, k-th launch of that kernel inside a step, grid) = per layer: calls, avg/min us, share of GPU time
  profiles/<tag>_hbm_traffic_pmc.csv      FETCH_SIZE / WRITE_SIZE per instance (KB, averaged over dispatches)
  profiles/<tag>_hbm_traffic.json         HBM bytes / launch for the kernels bench.py prices: (2*FETCH_SIZE + WRITE_SIZE) * 1024
                                          (gfx950: FETCH_SIZE reports half of a wide coalesced stream; guide's correction)
  profiles/<tag>_roofline.json            per in-step kernel: SURVEY 8d's algorithmic bytes of the layer pass it carries (0 for the
                                          statistics / fold / reduce / head / optimiser launches), rocprofv3 avg us, fraction of
                                          8 TB/s, PMC traffic and its ratio to the algorithmic bytes; the step's totals and its tail
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
        # one optimiser launch per step: k_adam (eager: beta^t from the host), k_adam_dev (+ k_adam_tick) under graph capture
        steps = max([1] + [sum(1 for r in rows if short(r[name_col]) == m) for m in ("k_adam", "k_adam_dev", "k_adam_tick")])
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
                          ("bwd_stats layer4", "k_bwd_stats_bpc<2,4>"), ("btlnk_bwd_stats", "k_btlnk_bwd_stats<2>"),
                          ("btlnk_fwd", "k_btlnk_fwd_t"), ("first_bwd", "k_first_bwd"), ("first_moments", "k_first_moments"),
                          ("apply_next layer1", "k_layer_apply_next_bpc<0,2>"),
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

    # ---- per-kernel roofline table of the default stack's train step (B = 4096, 2-32-16-32-64, T V = 204, latent 16) --------------
    # a layer's forward reads its input once and writes its output once; its backward reads dOut and the saved input and
    # writes dIn (no dIn for layer 1); bottleneck forward reads U and writes z, backward reads U and dz and writes dU.
    B, TVB, chans, lat = 4096, 4 * 204, [2, 32, 16, 32, 64], 16
    fwd = {i: TVB * (chans[i] + chans[i + 1]) for i in range(4)}
    bwd = {i: TVB * (chans[i + 1] + chans[i] + (chans[i] if i else 0)) for i in range(4)}
    carries = {   # kernel prefix -> (what, 8d bytes per clip)
        "k_layer_apply_next_bpc<0,2>": ("forward layer 1 (+ statistics of layer 2)", fwd[0]),
        "k_layer_apply_next_bpc<2,1>": ("forward layer 2 (+ statistics of layer 3)", fwd[1]),
        "k_layer_apply_next_bpc<1,2>": ("forward layer 3 (+ statistics of layer 4)", fwd[2]),
        "k_layer_apply_bpc<2>": ("forward layer 4", fwd[3]),
        "k_btlnk_fwd_t": ("bottleneck forward", 4 * 64 * 204 + 4 * lat),
        "k_btlnk_bwd_stats<2>": ("bottleneck backward + batch reductions of layer 4", 2 * 4 * 64 * 204 + 4 * lat),
        "k_btlnk_bwd": ("bottleneck backward", 2 * 4 * 64 * 204 + 4 * lat),
        "k_layer_bwd_bpc<2,4": ("backward layer 4 (+ batch reductions of layer 3)", bwd[3]),
        "k_layer_bwd_bpc<1,2": ("backward layer 3 (+ batch reductions of layer 2)", bwd[2]),
        "k_layer_bwd_bpc<2,1": ("backward layer 2 (+ batch reductions of layer 1)", bwd[1]),
        "k_first_bwd": ("backward layer 1", bwd[0]),
    }
    table, step_us, tail_us, tail_n = [], 0.0, 0.0, 0
    for (k, slot, wg), d in rows:
        if slot < 0 or not k.startswith("k_"):
            continue                              # not one launch per step: eval forwards, set-up copies
        avg = sum(d) / len(d)
        what, bpc = next(((w, b) for pre, (w, b) in carries.items() if k.startswith(pre)), ("statistics / fold / partial sums / head / optimiser", 0))
        byts = B * bpc
        tr = summary.get((k, slot, wg))
        table.append({"kernel": k, "launch_slot_in_step": slot, "workgroups": wg, "carries": what, "bytes_8d": byts, "avg_us": round(avg, 1),
                      "frac_of_8TBs": round(byts / (avg * 1e-6) / 8e12, 4) if byts else 0.0,
                      "traffic_bytes": int(tr[2]) if tr else None,
                      "traffic_ratio": round(tr[2] / byts, 2) if (tr and byts) else None})
        step_us += avg
        if avg < 25.0:
            tail_us += avg
            tail_n += 1
    total_8d = sum(r["bytes_8d"] for r in table)
    roof = {"note": "rocprofv3 --kernel-trace averages per in-step kernel instance (tools/collect_profiles.sh: python bench.py --steps 10, "
                    "B = 4096); bytes_8d = SURVEY 8d's algorithmic bytes of the layer pass the kernel carries x 4096 clips (0: a launch "
                    "8d does not count); traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from the PMC passes",
            "step": {"sum_of_kernel_avgs_us": round(step_us, 1), "bytes_8d": total_8d, "bytes_8d_per_clip": total_8d // B,
                     "frac_of_8TBs": round(total_8d / (step_us * 1e-6) / 8e12, 4)},
            "tail": {"launches_under_25us": tail_n, "their_sum_us": round(tail_us, 1)},
            "kernels": sorted(table, key=lambda r: -r["avg_us"])}
    with open(f"{dst}/{tag}_roofline.json", "w") as f:
        json.dump(roof, f, indent=1)
    print(open(f"{dst}/{tag}_kernel_instances.csv").read())
    print(json.dumps(out, indent=1))
    print(json.dumps({k: roof[k] for k in ("step", "tail")}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
