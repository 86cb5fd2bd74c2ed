#!/usr/bin/env python3
"""The floor under a LONE frame of the Julia pipeline (the reference's own call pattern: one frame per draw,
graphics.rs:324) -- VERDICT r03 item 6 / weak 2.

A lone 1080p frame is latency bound: its run time is the frame's longest ray, a chain of dependent march steps that one
wave issues back to back while most of the device idles.  The floor of that chain is

    max over rays [ steps inside the bounding sphere x I_in  +  orbit trips x I_trip  +  steps outside x I_out ]
        x  the issue interval of a wave alone on its SIMD  /  the clock

  * the per-ray counts come from the instrumented oracle (oracle.render_ray_costs: scene_SDF calls of the march, those
    past julia.wgsl:7-10's early-out, Julia iterations), every pose of the workload's orbit that is sampled;
  * I_in, I_trip, I_out are COUNTED from the source of the hand-written long-ray loop (kifs_julia_march_asm.hpp and the
    macros it expands, kifs_scene.hpp) -- the path a step takes through its labels, instruction by instruction, in the
    latency kernel's build (packed orbit trip, exit test every third trip, the short divide / square root when
    sdf_iters <= 24);
  * the issue interval is tools/microbench/issue_cost's figure for dependent instructions of a lone wave (4.89 cycles;
    gpurun_out/issue_cost.txt, profiles/r02/README.md), the clock 2.4 GHz (GRBM_GUI_ACTIVE / kernel time of the
    profiled launches).

It is a floor: taken branches, instruction fetch after a branch, the wave's set-up, the shading of the hit and the
store are not in it, and the kernel cannot go below it without fewer instructions on that ray's path.  Writes
profiles/lone_frame_floor.json (keyed by workload, stamped with the kernel-source hash); bench.py reports
secondary.lone_frame.floor_ms / frac_of_floor from it.  CPU only.

    python tools/lone_frame_floor.py [workload ...]        (default: the Julia workloads of bench.py's line)
"""
import json
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CSRC = ROOT / "kifs_raymarching_amd" / "csrc"
ISSUE_CYCLES = 4.89   # lone wave, dependent v_fma_f32 (tools/microbench/issue_cost)
CLOCK_HZ = 2.4e9
INSTR = re.compile(r'^\s*"\s*(v_|s_|ds_|global_|buffer_|flat_)')


def _macro(text, name):
    """The string-literal lines of `#define name ...` (continued with backslashes)."""
    start = text.index("#define " + name)
    lines = []
    for line in text[start:].split("\n")[1:]:
        lines.append(line)
        if not line.rstrip().endswith("\\"):
            break
    return lines


def _count(lines):
    return sum(1 for l in lines if INSTR.match(l))


def instruction_counts(short_divsqrt=True):
    """Instructions on the path of one march step through the labels of the asm statement, latency build."""
    scene = (CSRC / "kifs_scene.hpp").read_text()
    asm = (CSRC / "kifs_julia_march_asm.hpp").read_text().split("\n")
    trip = _count(_macro(scene, "KIFS_FAST_TRIP_PACKED"))
    prologue = _count(_macro(scene, "KIFS_JULIA_PROLOGUE_PACKED"))
    divsqrt = _count(_macro(scene, "KIFS_DIVSQRT_ORDINARY" if short_divsqrt else "KIFS_DIVSQRT_FULL"))

    def segment(a, b):
        """Instruction lines between the line holding `a` and the line holding `b` (exclusive)."""
        # macros: code lines that START with the name (a comment may follow), not mentions inside comments
        hit = lambda pat, l: bool(re.match(re.escape(pat) + r"(\s|$)", l.strip())) if pat.startswith("KIFS_") else (pat in l)
        ia = next(i for i, l in enumerate(asm) if hit(a, l))
        ib = next(i for i, l in enumerate(asm) if hit(b, l) and i > ia)
        return _count(asm[ia + 1:ib])
    head = segment('"10:\\n"', "KIFS_JULIA_PROLOGUE")            # dot(p,p), bound test, exec, branch
    enter = segment("KIFS_JULIA_PROLOGUE", "KIFS_ORBIT_LOOP")   # exec save, trip counters, remainder branch
    # the latency kernels' loop over blocks of six trips (KIFS_ORBIT_LOOP_PAIRED, kifs_scene.hpp): two blocks per pass
    paired = _macro(scene, "KIFS_ORBIT_LOOP_PAIRED")
    assert sum("KIFS_ORBIT_BLOCK_(KIFS_TRIP_EXIT)" in l for l in paired) == 4, "two blocks of six trips (four halves) per pass"
    i13 = next(i for i, l in enumerate(paired) if '"13:' in l)
    loop_head = _count(paired[:i13])                            # s_cmp, s_cbranch in front of the loop
    exit_tests = 2                                              # the two s_cbranch_execz of a block (every third trip)
    after_blocks = _count(paired[i13:])                         # behind the pair of blocks: 3 for the back edge + 2 for the odd block
    # (the blocks' own lines start with the block macro, not with a quote: they are counted through `trip` and `exit_tests`)
    loop_tail = 3
    assert after_blocks == loop_tail + 2, after_blocks
    rem_each = 3                                                # s_sub, s_cmp, s_cbranch per remainder trip (label 31)
    after = segment('"14:\\n"', "KIFS_JULIA_DIVSQRT")           # exec restore, class test, log
    finish = segment("KIFS_JULIA_DIVSQRT", '"41:\\n"')          # d = 0.25 lg * root; or-exec, outside test
    advance = segment('"41:\\n"', '"s_cmp_lt_i32 %[trips], %[maxit]')  # hit test, t += d, p, t < max, counters, loop branch
    out_block = segment('"45:\\n"', '"s_cmp_lg_u64 s[74:75], 0') + 2   # outside lanes: bookkeeping + the culls-on test
    out_sqrt = segment('"s_cbranch_scc0 48f', '"49:\\n"')
    out_tail = segment('"49:\\n"', '"48:\\n"')                  # d = norm - 2, the moving-away cull, back to 41
    return {"trip": trip, "prologue": prologue, "divsqrt": divsqrt, "head": head, "enter": enter, "loop_head": loop_head,
            "loop_tail": loop_tail, "exit_tests_per_block": exit_tests, "remainder_overhead_per_trip": rem_each,
            "after_orbit": after, "finish": finish, "advance": advance, "outside_block": out_block,
            "outside_sqrt": out_sqrt, "outside_tail": out_tail}


def step_model(ic, sdf_iters):
    """(I_in without trips, I_trip, I_out): a step's instructions as fixed + per-trip parts: the orbit loop's own
    instructions (exit tests, counters, branches) are spread over the trips of a full-length orbit; a remainder trip
    (sdf_iters mod 6) carries 3 of its own.  The fixed parts do not depend on the ray."""
    rem = sdf_iters % 6
    blocks = sdf_iters // 6
    # two blocks per pass of the loop: 3 loop instructions per pair, 2 more when the loop is left (is a block left over?),
    # 1 for the odd block's way back; 2 exit tests per block; 3 per remainder trip
    loop = ic["exit_tests_per_block"] * blocks + ic["loop_tail"] * (blocks // 2) + (2 if blocks >= 2 else 0) + (blocks % 2)
    per_trip = ic["trip"] + (loop + ic["remainder_overhead_per_trip"] * rem) / max(1, sdf_iters)
    i_in = ic["head"] + ic["prologue"] + ic["enter"] + ic["loop_head"] + ic["after_orbit"] + ic["divsqrt"] + ic["finish"] + ic["advance"]
    i_out = ic["head"] + 3 + ic["outside_block"] + ic["outside_sqrt"] + ic["outside_tail"] + ic["advance"]
    return i_in, per_trip, i_out


def floor_for(key, poses=12):
    import oracle as O
    import kifs_raymarching_amd as K  # (host packing only)
    from kifs_raymarching_amd.configs import WORKLOADS, orbit_camera
    w = WORKLOADS[key]
    assert int(w.gui.fractal_group) == 1, "the floor is modelled for the Julia pipeline's hand-written loop"
    ic = instruction_counts(short_divsqrt=w.iters[0] <= 24)
    i_in, i_trip, i_out = step_model(ic, w.iters[0])
    ub = K.uniform_bytes
    s = O.from_bytes(O.Screen, ub(w.screen.into_buffer_data()))
    o = O.from_bytes(O.Options, ub(w.gui.into_buffer_data()))
    n_orbit = max(w.frames, 120)
    per_pose = []
    for k in list(range(0, n_orbit, max(1, n_orbit // poses)))[:poses] + ["fixed"]:
        cam = w.camera if k == "fixed" else orbit_camera(w, k)
        c = O.from_bytes(O.Camera, ub(cam.into_buffer_data()))
        calls, inside, inner = O.render_ray_costs(s, c, o, O.iters(*w.iters))
        cost = inside.astype(np.float64) * i_in + inner.astype(np.float64) * i_trip + (calls.astype(np.float64) - inside) * i_out
        at = np.unravel_index(int(np.argmax(cost)), cost.shape)
        per_pose.append({"pose": k, "critical_ray": {"pixel": [int(at[1]), int(at[0])], "steps": int(calls[at]),
                                                      "steps_inside": int(inside[at]), "orbit_trips": int(inner[at])},
                         "instructions": int(round(float(cost[at]))),
                         "floor_ms": round(float(cost[at]) * ISSUE_CYCLES / CLOCK_HZ * 1e3, 5)})
    orbit = [p for p in per_pose if p["pose"] != "fixed"]
    return {"workload": key, "instructions_per_step": {"inside_fixed": i_in, "per_orbit_trip": round(i_trip, 3), "outside": i_out},
            "counted": ic, "issue_cycles": ISSUE_CYCLES, "clock_hz": CLOCK_HZ,
            "floor_ms_orbit_mean": round(float(np.mean([p["floor_ms"] for p in orbit])), 5),
            "floor_ms_fixed_view": next(p["floor_ms"] for p in per_pose if p["pose"] == "fixed"),
            "poses": per_pose}


def main():
    spec_path = ROOT / "kifs_raymarching_amd" / "build.py"
    import importlib.util
    spec = importlib.util.spec_from_file_location("_kb", spec_path)
    kb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kb)
    keys = sys.argv[1:] or ["cfg2_julia_1080p", "cfg4_julia_4096", "ref_julia_1080p", "cfg1_julia_256"]
    out_path = ROOT / "profiles" / "lone_frame_floor.json"
    res = json.loads(out_path.read_text()) if out_path.exists() else {}
    for k in keys:
        r = floor_for(k, poses=12 if "4096" not in k else 4)
        r["kernel_hash"] = kb.kernel_hash()
        res[k] = r
        print(k, "instructions per step:", r["instructions_per_step"], "| floor (orbit mean)", r["floor_ms_orbit_mean"], "ms, fixed view",
              r["floor_ms_fixed_view"], "ms", flush=True)
    out_path.write_text(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
