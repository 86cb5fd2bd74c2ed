#!/usr/bin/env python3
"""Condenses gpurun_out/prof_wl (tools/profile_workloads.sh) into profiles/rNN/workloads_pmc.json, one
kernel-stats CSV per workload under profiles/rNN/workloads/, a Markdown table on stdout, and refreshes
profiles/pmc_traffic.json (the file bench.py reads `roofline.traffic` and `roofline.valu_issue` from).

    python tools/profile_workloads_summary.py r03

Per workload@B (B frames per launch), for the render kernel of the pass:
  kernel_us               rocprofv3 --kernel-trace --stats average duration
  gpixel_s                B * W * H / kernel_us
  hbm_frac                4 B per pixel / kernel time / 8 TB/s
  traffic_over_alg        (WRITE_SIZE KiB + 2 x FETCH_SIZE KiB) / algorithmic bytes   (gfx950 corrections of
                          MI355X_MICROARCH.md's HBM section: FETCH_SIZE doubled)
  valu_per_pixel          SQ_INSTS_VALU (wave instructions) * 64 / pixels: lane-slots of vector work per pixel
  lanes_live              SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU
  waves_per_simd          4 * SQ_WAVE_CYCLES / (1024 * cycles)   (SQ_WAVE_CYCLES counts quad-cycles)
  valu_issue_frac         SQ_INSTS_VALU * 2.25 / (1024 * cycles): the vector pipes' busy share at the plain wave64
                          rate (tools/microbench/valu_rate); compares / packed / transcendental ones cost more
  salu_per_valu           SQ_INSTS_SALU / SQ_INSTS_VALU
"""
import csv
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
SRC = ROOT / "gpurun_out" / "prof_wl"
SIMDS = 1024



def profiled_kernel_hash(src_dir):
    """The kernel hash (kifs_raymarching_amd/build.py: kernel_hash()) of the library the passes ran: third line of the
    stamp the profiling script copied beside its output; "" when the run predates it."""
    p = Path(src_dir) / "srchash.txt"
    lines = p.read_text().split() if p.exists() else []
    return lines[2] if len(lines) > 2 else ""

def find_csv(d, suffix):
    hits = sorted(Path(d).rglob(f"*{suffix}"))
    return hits[0] if hits else None


def dominant_kernel(stats_csv):
    with open(stats_csv) as f:
        rows = [r for r in csv.DictReader(f) if "render_" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    return rows[0] if rows else None


def per_dispatch(path, kernel_prefix):
    acc = defaultdict(lambda: defaultdict(float))
    meta = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if not r["Kernel_Name"].startswith(kernel_prefix):
                continue
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            meta = {"vgprs": int(r["VGPR_Count"]), "sgprs": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                    "workgroup": int(r["Workgroup_Size"]), "grid": int(r["Grid_Size"])}
    out = {}
    for k, v in acc.items():
        vals = list(v.values())
        vals = vals[len(vals) // 3:]  # drop the warm-up dispatches
        out[k] = sum(vals) / max(1, len(vals))
    return out, meta



def same_run(src, d):
    """Is directory `d` from the run whose stamp lies in `src` (gpurun merges a call's files INTO gpurun_out/: the
    directories of earlier rounds' runs stay where they were)?"""
    a, b = Path(src) / "srchash.txt", Path(d) / "srchash.txt"
    return a.exists() and b.exists() and a.read_text() == b.read_text()

def main():
    from kifs_raymarching_amd.configs import WORKLOADS  # (imports the package: needs the built library)
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    dst = ROOT / "profiles" / rnd
    (dst / "workloads").mkdir(parents=True, exist_ok=True)
    traffic_path = ROOT / "profiles" / "pmc_traffic.json"
    traffic = json.loads(traffic_path.read_text()) if traffic_path.exists() else {}
    summary = {}
    for d in sorted(p for p in SRC.iterdir() if p.is_dir() and "@" in p.name):
        if not same_run(SRC, d):
            print(f"(skipped {d.name}: left over from another run)", file=sys.stderr)
            continue
        name, B = d.name.split("@")
        B = int(B)
        w = WORKLOADS[name]
        stats = find_csv(d / "trace", "kernel_stats.csv")
        if not stats:
            continue
        shutil.copy(stats, dst / "workloads" / f"{d.name}_kernel_stats.csv")
        dom = dominant_kernel(stats)
        if not dom:
            continue
        kname = dom["Name"].split("(")[0]
        rec = {"kernel": kname, "frames_per_launch": B, "launches": int(dom["Calls"]),
               "kernel_us": round(float(dom["AverageNs"]) / 1e3, 2),
               "kernel_us_min": round(float(dom["MinNs"]) / 1e3, 2), "kernel_us_max": round(float(dom["MaxNs"]) / 1e3, 2)}
        pixels = B * w.pixels
        rec["gpixel_s"] = round(pixels / (rec["kernel_us"] * 1e-6) / 1e9, 2)
        rec["hbm_frac"] = round(4.0 * pixels / (rec["kernel_us"] * 1e-6) / 8e12, 5)
        counters = {}
        for kind in ("sq", "write", "fetch"):
            path = find_csv(d / f"pmc_{kind}", "counter_collection.csv")
            if path:
                c, meta = per_dispatch(path, kname)
                counters.update(c)
                if meta:
                    rec.update(meta)
        rec["counters_per_launch"] = {k: round(v, 1) for k, v in sorted(counters.items())}
        if "WRITE_SIZE" in counters and "FETCH_SIZE" in counters:
            hbm = counters["WRITE_SIZE"] * 1024 + 2 * counters["FETCH_SIZE"] * 1024
            rec["hbm_bytes_per_launch"] = int(hbm)
            rec["traffic_over_alg"] = round(hbm / (4.0 * pixels), 4)
        if counters.get("GRBM_GUI_ACTIVE"):
            cycles = counters["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            rec["kernel_cycles"] = int(cycles)
            if counters.get("SQ_WAVE_CYCLES"):
                rec["waves_per_simd"] = round(4.0 * counters["SQ_WAVE_CYCLES"] / (SIMDS * cycles), 2)
            if counters.get("SQ_INSTS_VALU"):
                rec["valu_issue_frac"] = round(counters["SQ_INSTS_VALU"] * 2.25 / (SIMDS * cycles), 4)
        if counters.get("SQ_INSTS_VALU"):
            rec["valu_per_pixel"] = round(counters["SQ_INSTS_VALU"] * 64.0 / pixels, 1)
            if counters.get("SQ_THREAD_CYCLES_VALU"):
                rec["lanes_live"] = round(counters["SQ_THREAD_CYCLES_VALU"] / counters["SQ_INSTS_VALU"], 1)
            if counters.get("SQ_INSTS_SALU"):
                rec["salu_per_valu"] = round(counters["SQ_INSTS_SALU"] / counters["SQ_INSTS_VALU"], 3)
        summary[d.name] = rec
        if "hbm_bytes_per_launch" in rec:
            key = name + (f"@{B}" if B > 1 else "")
            traffic[key] = {"kernel_hash": profiled_kernel_hash(SRC),
                            "write_size_kib": round(counters["WRITE_SIZE"], 2), "fetch_size_kib": round(counters["FETCH_SIZE"], 2),
                            "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "kernel": kname, "camera": "orbit", "round": rnd}
            if counters.get("SQ_INSTS_VALU") and rec.get("kernel_cycles"):
                traffic[key]["valu_instructions_per_launch"] = int(counters["SQ_INSTS_VALU"])
                traffic[key]["kernel_cycles"] = rec["kernel_cycles"]
    (dst / "workloads_pmc.json").write_text(json.dumps(summary, indent=1) + "\n")
    traffic_path.write_text(json.dumps(traffic, indent=1) + "\n")
    cols = ["kernel_us", "gpixel_s", "hbm_frac", "traffic_over_alg", "valu_per_pixel", "lanes_live", "waves_per_simd",
            "valu_issue_frac", "salu_per_valu", "vgprs", "sgprs", "lds_bytes"]
    print("| workload@B | kernel | " + " | ".join(cols) + " |")
    print("|---|---|" + "---|" * len(cols))
    for k, r in summary.items():
        print(f"| {k} | `{r['kernel'].replace('void kifs::', '')}` | " + " | ".join(str(r.get(c, "")) for c in cols) + " |")


if __name__ == "__main__":
    main()
