#!/usr/bin/env python3
"""Condenses gpurun_out/prof (tools/profile_round.sh) into the files kept under profiles/rNN/ and
refreshes profiles/pmc_traffic.json.

    python tools/profile_summary.py r02

Per render kernel and launch shape: rocprofv3's kernel-trace statistics, the per-dispatch PMC rows
of the render kernel (first rows kept verbatim), and derived figures -- HBM bytes per launch
(WRITE_SIZE in KiB as reported; FETCH_SIZE doubled, the gfx950 correction of MI355X_MICROARCH.md),
average resident waves per SIMD (SQ_WAVE_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES per SE...)), lanes
live per VALU instruction."""
import csv
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "prof"
SIMDS = 1024  # 256 CUs x 4



def profiled_kernel_hash(src_dir):
    """The kernel hash (kifs_raymarching_amd/build.py: kernel_hash()) of the library the passes ran: third line of the
    stamp the profiling script copied beside its output; "" when the run predates it."""
    p = Path(src_dir) / "srchash.txt"
    lines = p.read_text().split() if p.exists() else []
    return lines[2] if len(lines) > 2 else ""

def find_csv(d, suffix):
    hits = sorted(Path(d).rglob(f"*{suffix}"))
    return hits[0] if hits else None


def render_rows(path):
    with open(path) as f:
        return [r for r in csv.DictReader(f) if "render_" in r["Kernel_Name"]]


def per_dispatch(rows):
    """{counter: [value per dispatch]} (a dispatch's counter may be split over several rows: summed)."""
    acc = defaultdict(lambda: defaultdict(float))
    for r in rows:
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in acc.items()}


def mean(xs):
    return sum(xs) / len(xs) if xs else 0.0


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    dst = ROOT / "profiles" / rnd
    dst.mkdir(parents=True, exist_ok=True)
    for name in ("cfg2_bench.json", "other_workloads_bench.jsonl"):
        if (SRC / name).exists():
            shutil.copy(SRC / name, dst / name)
    stats = find_csv(SRC / "trace", "kernel_stats.csv")
    if stats:
        shutil.copy(stats, dst / "cfg2_kernel_stats.csv")
    traffic_path = ROOT / "profiles" / "pmc_traffic.json"
    traffic = json.loads(traffic_path.read_text()) if traffic_path.exists() else {}
    summary = {}
    for tag, B in (("", 48), ("_b8", 8), ("_b1", 1)):
        rec = {}
        keep = []
        for kind in ("write", "fetch", "sq"):
            path = find_csv(SRC / f"pmc_{kind}{tag}", "counter_collection.csv")
            if not path:
                continue
            rows = render_rows(path)
            if not rows:
                continue
            rec["kernel"] = rows[-1]["Kernel_Name"].split("(")[0]
            rec["vgprs"], rec["sgprs"], rec["lds_bytes"] = rows[-1]["VGPR_Count"], rows[-1]["SGPR_Count"], rows[-1]["LDS_Block_Size"]
            # drop the warm-up dispatches (first third) before averaging
            pd = per_dispatch(rows)
            for k, v in pd.items():
                rec[k] = mean(v[len(v) // 3:])
            keep += rows[-24:]
        if not rec:
            continue
        if "WRITE_SIZE" in rec and "FETCH_SIZE" in rec:
            rec["hbm_bytes_per_launch"] = int(rec["WRITE_SIZE"] * 1024 + 2 * rec["FETCH_SIZE"] * 1024)
            key = "cfg2_julia_1080p" + (f"@{B}" if B > 1 else "")
            traffic[key] = {"kernel_hash": profiled_kernel_hash(SRC),
                            "write_size_kib": round(rec["WRITE_SIZE"], 2), "fetch_size_kib": round(rec["FETCH_SIZE"], 2),
                            "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "kernel": rec["kernel"],
                            "camera": "orbit", "round": rnd}
            if rec.get("SQ_INSTS_VALU") and rec.get("GRBM_GUI_ACTIVE"):
                traffic[key]["valu_instructions_per_launch"] = int(rec["SQ_INSTS_VALU"])
                traffic[key]["kernel_cycles"] = int(rec["GRBM_GUI_ACTIVE"] / 8.0)
        if rec.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = the launch's duration in shader cycles
            cycles = rec["GRBM_GUI_ACTIVE"] / 8.0
            rec["kernel_cycles"] = cycles
            if "SQ_WAVE_CYCLES" in rec:
                # SQ_WAVE_CYCLES counts QUAD-cycles (MI355X_MICROARCH.md, s_memtime tick vs SQ PMC units).
                # `raw` is the ratio round 1's VERDICT quoted (1.1 then); x4 = resident waves per SIMD,
                # barrier-parked waves included
                rec["wave_cycles_per_simd_cycle_raw"] = rec["SQ_WAVE_CYCLES"] / (SIMDS * cycles)
                rec["resident_waves_per_simd"] = 4.0 * rec["SQ_WAVE_CYCLES"] / (SIMDS * cycles)
            if rec.get("SQ_INSTS_VALU"):
                # a wave64 VALU instruction takes the SIMD-32 two cycles (packed-f32 ones four: the orbit's)
                rec["valu_issue_busy_at_2_cycles"] = rec["SQ_INSTS_VALU"] * 2.0 / (SIMDS * cycles)
        if rec.get("SQ_INSTS_VALU") and rec.get("SQ_THREAD_CYCLES_VALU"):
            rec["lanes_live_per_valu"] = rec["SQ_THREAD_CYCLES_VALU"] / rec["SQ_INSTS_VALU"]
        summary[f"frames_per_launch_{B}"] = rec
        with open(dst / f"cfg2_pmc_rows{tag or '_b48'}.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
            w.writeheader()
            w.writerows(keep)
    (dst / "cfg2_pmc_summary.json").write_text(json.dumps(summary, indent=1) + "\n")
    traffic["_note"] = ("HBM bytes per launch of the render kernel from rocprofv3 --pmc passes (separate WRITE_SIZE and "
                        "FETCH_SIZE runs of `bench.py --frames-per-launch B`, tools/profile_round.sh; rows kept under "
                        "profiles/<round>/cfg2_pmc_rows_b*.csv). WRITE_SIZE is in KiB (the store pattern -- 4 B/lane in full "
                        "128-B row segments -- self-calibrates because the written byte count is known); FETCH_SIZE is doubled "
                        "per the gfx950 correction in MI355X_MICROARCH.md (HBM section). Keys: workload for one frame per "
                        "launch, workload@B for launches of B frames.")
    traffic_path.write_text(json.dumps(traffic, indent=1) + "\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
