"""Condense a scripts/profile_round.sh output directory into the small text/JSON summaries that
are committed under profiles/ (the raw rocprofv3 CSVs stay in gpurun_out/)."""
import collections
import csv
import glob
import json
import os
import sys


def main(out, rnd, var):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "gpurun_out", "profiles_%s" % rnd)
    os.makedirs(dst, exist_ok=True)
    cmdline = ("--steps 20 --warmup 3 --no-cpu --no-configs --sustain 0 --variant %s" % var) if "_" not in var else \
        ("--config %s --variant %s%s --steps %s --warmup 1 --no-cpu" % (tuple(var.split("_", 1)) + (
            (" --batch " + os.environ["TC_PROFILE_BATCH"]) if int(os.environ.get("TC_PROFILE_BATCH", "0")) else "",
            os.environ.get("TC_PROFILE_STEPS", "3"))))
    lines = ["rocprofv3 --kernel-trace --stats summary of: python3 bench.py " + cmdline,
             "Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev"]
    kavg = None
    newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1:]      # noqa: E731  (earlier runs leave their files behind)
    for f in newest(os.path.join(out, "kt", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            nm = r["Name"]
            if len(nm) > 90:
                nm = nm[:60] + "...<%d chars>" % len(nm)
            lines.append(",".join([nm, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                                   r["MinNs"], r["MaxNs"], r["StdDev"]]))
            if r["Name"].startswith("tc_solve_kernel"):
                kavg = float(r["AverageNs"])
    pm = {}
    info = {}
    for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        for f in newest(os.path.join(out, d, "*", "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith("tc_solve_kernel"):
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    info = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size",
                                              "LDS_Block_Size", "Grid_Size", "Workgroup_Size") if k in r}
            for k, v in agg.items():
                pm[k] = sum(v) / len(v)
    lines.append("")
    lines.append("PMC counters, tc_solve_kernel, average per launch (separate --pmc passes):")
    for k in sorted(pm):
        lines.append("%s,%.6g" % (k, pm[k]))
    lines.append("dispatch info: " + json.dumps(info))
    traffic = None
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM)
        traffic = (2 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024
        lines.append("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 = %.6g" % traffic)
    with open(os.path.join(dst, "%s_%s_rocprof_summary.txt" % (rnd, var)), "w") as f:
        f.write("\n".join(lines) + "\n")
    tj = os.path.join(dst, "traffic.json")
    t = json.load(open(tj)) if os.path.exists(tj) else {}
    if traffic is not None:
        B = {"quadcopter": 4096, "l1l2": 16384, "dcmotor": 32768}.get(var.split("_")[0], 65536)
        B = int(os.environ.get("TC_PROFILE_BATCH", "0")) or B
        t["%s_B%d" % (var, B)] = {"hbm_bytes_per_launch": traffic, "fetch_size_kib": pm["FETCH_SIZE"],
                                "write_size_kib": pm["WRITE_SIZE"], "kernel_avg_ns": kavg, "round": rnd}
        if "SQ_INSTS_VALU" in pm:       # wave-level VALU instructions per launch: bench.py's lane-redundancy ratio
            t["%s_B%d" % (var, B)]["sq_insts_valu"] = pm["SQ_INSTS_VALU"]
    json.dump(t, open(tj, "w"), indent=1)
    print("\n".join(lines[-14:]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
