#!/usr/bin/env python3
"""Condenses gpurun_out/prof_stalls (tools/profile_stalls.sh) into profiles/rNN/stalls.json and a Markdown table on
stdout: per workload@B, for the render kernel of the launch, where a resident wave's cycles go.

    python tools/profile_stalls_summary.py r04

SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, "rocprofv3 PMC
slots"): WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (an instruction is ready but the issue port or its
pipe is not: the arbiter picked another wave, or the pipe is busy) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES.  Columns:

  waves_per_simd     4 x SQ_WAVE_CYCLES / (1024 SIMDs x kernel cycles)
  active / wait_inst / wait_any   shares of SQ_WAVE_CYCLES
  wait_lds           SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES (a part of wait_inst)
  valu / sca / lds / misc         SQ_ACTIVE_INST_{VALU,SCA,LDS,MISC} / SQ_ACTIVE_INST_ANY: what the issuing cycles issue
  simd_valu_busy     4 x SQ_ACTIVE_INST_VALU / (1024 x kernel cycles): the share of all SIMD-cycles in which a vector
                     instruction of SOME wave occupies the pipe -- the figure to hold against valu_issue_frac
                     (instructions x 2.25 cycles), which prices every instruction at the plain rate
  per VALU instruction: salu, lds, smem, branch instruction counts; lanes live; cycles (4 x ACTIVE_INST_VALU / INSTS_VALU)
  lds_conflict       SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "prof_stalls"
SIMDS = 1024


def per_dispatch(path):
    """Mean per launch of every counter over the later two thirds of the dispatches of the dominant render kernel."""
    acc = defaultdict(lambda: defaultdict(float))
    meta = {}
    names = defaultdict(int)
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        if "render_" in r["Kernel_Name"]:
            names[r["Kernel_Name"]] += int(r["Grid_Size"])
    if not names:
        return {}, {}
    kernel = max(names, key=names.get)
    for r in rows:
        if r["Kernel_Name"] != kernel:
            continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        meta = {"kernel": kernel.replace("void kifs::", "").split("(")[0], "vgpr_granules": int(r["VGPR_Count"]),
                "sgprs": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]), "workgroup": int(r["Workgroup_Size"]),
                "grid": int(r["Grid_Size"])}
    out = {}
    for k, v in acc.items():
        vals = [v[d] for d in sorted(v, key=int)]
        vals = vals[len(vals) // 3:]
        out[k] = sum(vals) / max(1, len(vals))
    return out, meta



def same_run(src, d):
    """Is directory `d` from the run whose stamp lies in `src` (gpurun merges a call's files INTO gpurun_out/: the
    directories of earlier rounds' runs stay where they were)?"""
    a, b = Path(src) / "srchash.txt", Path(d) / "srchash.txt"
    return a.exists() and b.exists() and a.read_text() == b.read_text()

def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    dst = ROOT / "profiles" / rnd
    dst.mkdir(parents=True, exist_ok=True)
    summary = {}
    for d in sorted(p for p in SRC.iterdir() if p.is_dir() and "@" in p.name):
        if not same_run(SRC, d):
            print(f"(skipped {d.name}: left over from another run)", file=sys.stderr)
            continue
        c, meta = {}, {}
        for p in ("pmc_a", "pmc_b", "pmc_c"):
            hits = sorted((d / p).rglob("*counter_collection.csv"))
            if hits:
                cc, mm = per_dispatch(hits[0])
                c.update(cc)
                meta = mm or meta
        if not c.get("SQ_WAVE_CYCLES"):
            continue
        wc = c["SQ_WAVE_CYCLES"]
        cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
        any_ = c.get("SQ_ACTIVE_INST_ANY", 0.0)
        valu = c.get("SQ_INSTS_VALU", 0.0)
        r = dict(meta)
        r["kernel_cycles"] = int(cycles)
        r["waves_per_simd"] = round(4.0 * wc / (SIMDS * cycles), 2) if cycles else None
        r["active"] = round(any_ / wc, 3)
        r["wait_inst"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
        r["wait_any"] = round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3)
        r["wait_lds"] = round(c.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4)
        for k, n in (("valu", "SQ_ACTIVE_INST_VALU"), ("sca", "SQ_ACTIVE_INST_SCA"), ("lds", "SQ_ACTIVE_INST_LDS"),
                     ("misc", "SQ_ACTIVE_INST_MISC")):
            r["issue_" + k] = round(c.get(n, 0.0) / any_, 3) if any_ else None
        if cycles:
            r["simd_valu_busy"] = round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (SIMDS * cycles), 3)
            r["valu_issue_frac_plain"] = round(valu * 2.25 / (SIMDS * cycles), 3)
            r["simd_busy"] = round(4.0 * c.get("SQ_BUSY_CYCLES", 0.0) / (8 * 4 * cycles), 3) if c.get("SQ_BUSY_CYCLES") else None
        if valu:
            r["cycles_per_valu"] = round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / valu, 2)
            r["salu_per_valu"] = round(c.get("SQ_INSTS_SALU", 0.0) / valu, 3)
            r["lds_per_valu"] = round(c.get("SQ_INSTS_LDS", 0.0) / valu, 4)
            r["smem_per_valu"] = round(c.get("SQ_INSTS_SMEM", 0.0) / valu, 4)
            r["branch_per_valu"] = round(c.get("SQ_INSTS_BRANCH", 0.0) / valu, 4)
            r["lanes_live"] = round(c.get("SQ_THREAD_CYCLES_VALU", 0.0) / valu, 1) if c.get("SQ_THREAD_CYCLES_VALU") else None
        if c.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_conflict"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("SQ_INST_CYCLES_SALU") and c.get("SQ_INSTS_SALU"):
            r["cycles_per_salu"] = round(4.0 * c["SQ_INST_CYCLES_SALU"] / c["SQ_INSTS_SALU"], 2)
        r["counters_per_launch"] = {k: round(v, 1) for k, v in sorted(c.items())}
        summary[d.name] = r
    (dst / "stalls.json").write_text(json.dumps(summary, indent=1) + "\n")
    cols = ["waves_per_simd", "active", "wait_inst", "wait_any", "wait_lds", "issue_valu", "issue_sca", "issue_lds",
            "issue_misc", "simd_valu_busy", "valu_issue_frac_plain", "cycles_per_valu", "salu_per_valu", "lds_per_valu",
            "branch_per_valu", "lanes_live", "lds_conflict"]
    print("| workload@B | kernel | " + " | ".join(cols) + " |")
    print("|---|---|" + "---|" * len(cols))
    for k, r in summary.items():
        print(f"| {k} | `{r.get('kernel', '')}` | " + " | ".join(str(r.get(c, "")) for c in cols) + " |")


if __name__ == "__main__":
    main()
