"""Condense gpurun_out/<tag>/ (tools/profile_round3.sh) into the committed summaries under profiles/: <tag>_kernel_stats.csv,
<tag>_rocprofv3_summary.json, <tag>_bench_<workload>.json, and the `1m_team` / `256k_team` / `64k_team` entries of pmc_traffic.json,
each keyed to the kernel sources it was measured on.  python tools/collect_profiles3.py [tag]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
O = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")


def pmc_of(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(pattern, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "fftk::" not in k:
                continue
            k = k[k.index("fftk::"):k.index(">") + 1]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
    return {k: {c: val / cnt[(k, c)] for c, val in v.items()} for k, v in agg.items()}


def derive(c):
    d = {}
    if "FETCH_SIZE" in c:
        d["fetch_GB_x2"] = c["FETCH_SIZE"] * 2.048e-6   # KB units; 128-byte requests tallied at 64 (MI355X_MICROARCH.md, HBM)
    if "WRITE_SIZE" in c:
        d["write_GB"] = c["WRITE_SIZE"] * 1.024e-6
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0))
    if "SQ_LDS_BANK_CONFLICT" in c:
        d["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"])
        d["wait_any_frac"] = c["SQ_WAIT_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"])
        d["valu_active_frac"] = c["SQ_ACTIVE_INST_VALU"] / max(1.0, c["SQ_WAVE_CYCLES"])
    return d


def main():
    from bench import kernel_source_hash
    out = {"kernel_stats": [], "pmc": {}, "derived": {}, "kernel_source_hash": kernel_source_hash()}
    for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
        shutil.copy(f, os.path.join(P, "%s_kernel_stats.csv" % tag))
        for row in csv.DictReader(open(f)):
            if "fftk::" in row["Name"]:
                out["kernel_stats"].append({k: row[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
    for f in glob.glob(O + "/bench_*.json"):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(P, "%s_%s" % (tag, os.path.basename(f))), "w").write(lines[-1] + "\n")
    lines = [l for l in open(O + "/trace_bench.log").read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(P, "%s_bench_under_rocprof.json" % tag), "w").write(lines[-1] + "\n")
    pt_path = os.path.join(P, "pmc_traffic.json")
    pt = json.load(open(pt_path))
    for wl in ("1m", "256k", "64k"):
        pm = pmc_of(O + "/pmc_%s_*/**/*counter_collection.csv" % wl)
        team = [k for k in pm if "team_" in k]
        out["pmc"][wl] = {k: pm[k] for k in team}
        out["derived"][wl] = {k: derive(pm[k]) for k in team}
        bpath = os.path.join(P, "%s_bench_%s.json" % (tag, wl))
        if not team or not os.path.exists(bpath):
            continue
        b = json.loads(open(bpath).read())
        if b["config"]["team_status"] != 0:
            continue
        k = team[0]
        d = out["derived"][wl][k]
        alg = b["roofline"]["algorithmic_bytes_per_launch_set"]
        tot = (d.get("fetch_GB_x2", 0) + d.get("write_GB", 0)) * 1e9
        pt[wl + "_team"] = {
            "factors": b["config"]["factors"], "units_per_launch": b["roofline"]["units_per_launch_set"],
            "kernel": k, "kernel_source_hash": out["kernel_source_hash"],
            "hbm_bytes_per_launch_set": tot, "fetch_bytes_x2": d.get("fetch_GB_x2", 0) * 1e9, "write_bytes": d.get("write_GB", 0) * 1e9,
            "l2_hit_rate": d.get("l2_hit_rate"), "traffic_over_algorithmic": tot / alg,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in separate runs (tools/profile_round3.sh %s), per launch "
                    "of %d transforms of %s.  Bytes that cross the L2's memory-side interface (Infinity-Cache hits included); FETCH_SIZE "
                    "tallies 128-byte requests at 64 and is doubled (MI355X_MICROARCH.md, HBM section).  Reads %.2f GB, writes %.2f GB against "
                    "%.2f + %.2f GB algorithmic = %.2f x; L2 hit rate %.1f %%."
                    % (tag, b["roofline"]["units_per_launch_set"], k, d.get("fetch_GB_x2", 0), d.get("write_GB", 0), alg / 2e9, alg / 2e9, tot / alg,
                       100 * (d.get("l2_hit_rate") or 0))}
    json.dump(pt, open(pt_path, "w"), indent=1)
    json.dump(out, open(os.path.join(P, "%s_rocprofv3_summary.json" % tag), "w"), indent=1)
    print(json.dumps({"kernel_stats": out["kernel_stats"][:4], "derived": out["derived"]}, indent=1)[:3000])
    for f in sorted(glob.glob(os.path.join(P, "%s_bench_*.json" % tag))):
        r = json.loads(open(f).read())
        print(os.path.basename(f), "value %.1f (median %.1f) frac %.3f traffic %s" % (r["value"], r.get("value_median", 0), r["roofline"]["frac"], r["roofline"].get("traffic")),
              r["roofline"].get("copy_gbs"), r.get("secondary_fp64", {}).get("value") if r.get("secondary_fp64") else None)


if __name__ == "__main__":
    main()
