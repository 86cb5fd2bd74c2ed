"""The committed measurement records bench.py quotes from (profiles/pmc_traffic.json: HBM bytes and VALU instructions per
launch from rocprofv3 --pmc passes; profiles/lone_frame_floor.json: the lone frame's floor) are tied to the kernels they
were taken from by build.kernel_hash() -- bench.py reports them only for a library built from the same kernel sources.
CPU: the floor is re-derivable here (source + oracle), so it must be CURRENT; the counter record needs the GPU, so it may
lag, and then bench.py must say so rather than quote it (tests/test_gpu_bench.py checks the line)."""
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _kernel_hash():
    spec = importlib.util.spec_from_file_location("_kb", ROOT / "kifs_raymarching_amd" / "build.py")
    kb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kb)
    return kb.kernel_hash()


def test_the_lone_frame_floor_belongs_to_the_kernels_in_the_tree(kifs):
    rec = json.loads((ROOT / "profiles" / "lone_frame_floor.json").read_text())
    now = _kernel_hash()
    assert kifs._lib.kernel_hash_of_loaded_library() == now
    for key in ("cfg2_julia_1080p", "cfg4_julia_4096", "ref_julia_1080p"):
        assert rec[key]["kernel_hash"] == now, f"{key}: re-run tools/lone_frame_floor.py (the kernel sources changed)"
        assert 0 < rec[key]["floor_ms_fixed_view"] and 0 < rec[key]["floor_ms_orbit_mean"]
        ips = rec[key]["instructions_per_step"]
        assert 60 <= ips["inside_fixed"] <= 140 and 9 <= ips["per_orbit_trip"] <= 12 and 30 <= ips["outside"] <= 70


def test_the_floor_counts_what_the_source_says(oracle, kifs):
    sys.path.insert(0, str(ROOT / "tools"))
    import lone_frame_floor as LF
    ic = LF.instruction_counts(short_divsqrt=True)
    assert ic["trip"] == 9 and ic["prologue"] == 8   # 8 v_pk_*_f32 + v_cmpx; the squares of q_0
    assert ic["head"] == 7 and ic["enter"] == 4 and ic["loop_head"] == 2 and ic["advance"] == 11
    assert LF.instruction_counts(short_divsqrt=False)["divsqrt"] > ic["divsqrt"] > 20
    # cfg1 is small enough to recount here: the critical ray of the fixed view
    r = LF.floor_for("cfg1_julia_256", poses=1)
    fixed = next(p for p in r["poses"] if p["pose"] == "fixed")
    ray = fixed["critical_ray"]
    i_in, i_trip, i_out = (r["instructions_per_step"][k] for k in ("inside_fixed", "per_orbit_trip", "outside"))
    want = ray["steps_inside"] * i_in + ray["orbit_trips"] * i_trip + (ray["steps"] - ray["steps_inside"]) * i_out
    assert abs(fixed["instructions"] - want) <= 1 and ray["steps"] <= 64 and ray["orbit_trips"] <= 8 * ray["steps_inside"]
    assert fixed["floor_ms"] == round(fixed["instructions"] * LF.ISSUE_CYCLES / LF.CLOCK_HZ * 1e3, 5) or abs(
        fixed["floor_ms"] - want * LF.ISSUE_CYCLES / LF.CLOCK_HZ * 1e3) < 1e-4


def test_counter_records_name_their_kernels():
    rec = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    entries = {k: v for k, v in rec.items() if isinstance(v, dict)}
    assert "cfg2_julia_1080p@48" in entries
    for k, v in entries.items():
        assert "hbm_bytes_per_launch" in v and "kernel" in v, k
