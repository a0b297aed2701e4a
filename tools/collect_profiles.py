#!/usr/bin/env python3
"""Condense what tools/refresh_profiles.sh left under gpurun_out/ into profiles/ (tracked):
  <tag>_bench_{default,lanes1,tvl1_lanes1,deepflow_lanes1}_kernel_stats.csv + .json   rocprofv3 --kernel-trace --stats summaries and the lines those runs printed
  <tag>_bench_pmc.json                                      the line of `bench.py --pmc` (roofline.traffic measured in that run)
  hbm_traffic.json                                          FETCH_SIZE / WRITE_SIZE per launch of the dominant kernels, tagged with the kernel source fingerprint
  <tag>_sq_counters.json                                    SQ / GRBM counters of k_iter2_rows and k_df_sor_rt: totals, per-launch means, derived shares
usage: python tools/collect_profiles.py r03 [destination directory, default profiles/]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "profiles")
SIMDS = 1024


def sq_summary(tag, algo, kernel):
    base = os.path.join(ROOT, "gpurun_out", f"pmc_sq_{tag}_{algo}")
    passes = {}
    for f in sorted(glob.glob(os.path.join(base, "sq*", "sq*_counter_collection.csv"))):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                d = int(r["Dispatch_Id"])
                per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                per[d]["_grid"] = int(r["Grid_Size"])
                per[d]["_vgpr"] = int(r["VGPR_Count"]); per[d]["_sgpr"] = int(r["SGPR_Count"])
        passes[os.path.basename(os.path.dirname(f))] = [per[i] for i in sorted(per)]
    if not passes:
        return None
    n = min(len(v) for v in passes.values())
    gmax = max(p["_grid"] for v in passes.values() for p in v)
    tot, cnt = collections.defaultdict(float), collections.Counter()
    tot_all = collections.defaultdict(float)       # every dispatch of the pass (the pass profiles exactly the timed steps: --steps-only)
    rows = []
    for i in range(n):
        row = {}
        for v in passes.values():
            row.update({k: x for k, x in v[i].items() if not k.startswith("_")})
            row["_grid"] = v[i]["_grid"]
        rows.append(row)
        for k, x in row.items():
            if not k.startswith("_"):
                tot_all[k] += x
        if algo == "deepflow" and row["_grid"] < gmax // 8:
            continue                                           # (older runs: single-pair latency launches of the same kernel)
        for k, x in row.items():
            if not k.startswith("_"):
                tot[k] += x; cnt[k] += 1
    sys.path.insert(0, ROOT)
    from bench import kernel_source_fingerprint
    out = {"kernel": kernel, "source_fingerprint": kernel_source_fingerprint(), "dispatches": n, "command": f"tools/pmc_sq.sh {tag} {algo}  (rocprofv3 --pmc <<=5 counters per pass> --kernel-include-regex ... -- "
                                                           "python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile --steps-only --in-flight 1 --lanes 1 --no-deepflow --batch 128" + (" --algo deepflow)" if algo == "deepflow" else ")"),
           "units": "SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs",
           "totals": {k: tot[k] for k in sorted(tot)}, "per_dispatch_mean": {k: tot[k] / cnt[k] for k in sorted(tot)},
           "totals_all_dispatches": {k: tot_all[k] for k in sorted(tot_all)}}
    t = out["totals"]
    if "SQ_WAVE_CYCLES" in t:
        wc = t["SQ_WAVE_CYCLES"]
        d = {"wave_time_share_active": t.get("SQ_ACTIVE_INST_ANY", 0) / wc, "wave_time_share_issue_stalled": t.get("SQ_WAIT_INST_ANY", 0) / wc,
             "wave_time_share_parked_waitcnt_or_barrier": t.get("SQ_WAIT_ANY", 0) / wc, "wave_time_share_valu_active": t.get("SQ_ACTIVE_INST_VALU", 0) / wc}
        if "GRBM_GUI_ACTIVE" in t and "SQ_ACTIVE_INST_VALU" in t:
            cyc = t["GRBM_GUI_ACTIVE"] / 8.0
            d["valu_pipe_busy_per_simd_all_launches"] = t["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / cyc
            d["cycles_per_valu_inst_while_active"] = t["SQ_ACTIVE_INST_VALU"] * 4 / t["SQ_INSTS_VALU"]
            d["simd_cycles_per_valu_inst_all_launches"] = cyc * SIMDS / t["SQ_INSTS_VALU"]
            d["mean_waves_per_simd_resident"] = t["SQ_WAVE_CYCLES"] * 4 / SIMDS / cyc
        if "SQ_LDS_BANK_CONFLICT" in t and t.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share_of_lds_cycles"] = t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"]
        # the largest launches (all pairs iterating at full size): where the roofs are
        if all("SQ_INSTS_VALU" in r and "GRBM_GUI_ACTIVE" in r and "SQ_ACTIVE_INST_VALU" in r for r in rows):
            big = max(r["SQ_INSTS_VALU"] for r in rows)
            sel = [r for r in rows if r["SQ_INSTS_VALU"] >= 0.8 * big]
            d["full_launches"] = {"n": len(sel),
                                  "valu_pipe_busy_per_simd": sum(r["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / (r["GRBM_GUI_ACTIVE"] / 8) for r in sel) / len(sel),
                                  "simd_cycles_per_valu_inst": sum(r["GRBM_GUI_ACTIVE"] / 8 * SIMDS / r["SQ_INSTS_VALU"] for r in sel) / len(sel),
                                  "wave_time_share_parked": sum(r.get("SQ_WAIT_ANY", 0) / r["SQ_WAVE_CYCLES"] for r in sel) / len(sel),
                                  "wave_time_share_issue_stalled": sum(r.get("SQ_WAIT_INST_ANY", 0) / r["SQ_WAVE_CYCLES"] for r in sel) / len(sel)}
        out["derived"] = d
    return out


def main():
    global DST
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    if len(sys.argv) > 2:                       # on the GPU box: condense into a small directory under gpurun_out/ (profiles/ does not travel back)
        DST = sys.argv[2]
        os.makedirs(DST, exist_ok=True)
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    for name in ("default", "lanes1", "tvl1_lanes1", "deepflow_lanes1"):
        st = glob.glob(os.path.join(src, name, "**", "*kernel_stats.csv"), recursive=True)
        if st:
            shutil.copy(st[0], os.path.join(DST, f"{tag}_bench_{name}_kernel_stats.csv"))
        js = os.path.join(src, f"{tag}_bench_{name}.json")
        if os.path.exists(js) and os.path.getsize(js):
            shutil.copy(js, os.path.join(DST, f"{tag}_bench_{name}.json"))
    st = glob.glob(os.path.join(src, "saliency", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(DST, f"{tag}_saliency_kernel_stats.csv"))
    for name in (f"{tag}_bench_pmc.json", f"{tag}_launch_profile_b64.txt", f"{tag}_ablate_probe.txt", f"{tag}_saliency_bench.txt", f"{tag}_queue_forms.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(DST, name))
    ht = os.path.join(src, "pmc_live", "hbm_traffic.json")
    if os.path.exists(ht):
        shutil.copy(ht, os.path.join(DST, "hbm_traffic.json"))
    sq = {}
    for algo, kern in (("TVL1", "k_iter2_rows"), ("deepflow", "k_df_sor_rt")):
        s = sq_summary(tag, algo, kern)
        if s:
            sq[kern] = s
    if sq:
        with open(os.path.join(DST, f"{tag}_sq_counters.json"), "w") as f:
            json.dump(sq, f, indent=1)
        for k, v in sq.items():
            print(k, json.dumps(v.get("derived"), indent=1))
    for p in sorted(os.listdir(DST)):
        print(p)


if __name__ == "__main__":
    main()
