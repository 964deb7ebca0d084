#!/usr/bin/env python3
"""rocprofv3 passes over `bench.py --workload W` on the GPU box, summarised per kernel.

    python tools/profile_frame.py <workload> <out_dir> [pass ...]

Passes (each one rocprofv3 run of its own; counters never share a run with --kernel-trace):
  trace   --kernel-trace                       -> <out_dir>/<W>_kernel_stats.csv
  fp32    SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, SQ_INSTS_VALU, _SALU, _SMEM, _LDS
  wait    SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY,
          SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_SCA, SQ_WAVES
  lanes   SQ_THREAD_CYCLES_VALU, SQ_ACTIVE_INST_VALU, SQ_INST_CYCLES_VALU (lane_active_frac = thread-cycles / (64 x cycles))
  fetch   FETCH_SIZE          write   WRITE_SIZE
The counter passes end in <out_dir>/<W>_pmc.json: per kernel of the timed frames the mean per launch of
every counter (only kernels launched once per timed step count as the frame's), the hardware-counted FP32 operations per frame (64 lanes x (ADD + MUL + 2 FMA + TRANS)
wave-level instructions) and the HBM bytes per frame (FETCH_SIZE doubled: gfx950 tallies 128-byte
requests at 64 bytes; both counters are in KiB), stamped with the hash of the kernel sources they were
measured on (bench.py compares it with the sources it runs).
"""
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = {
    "fp32": ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
             "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS"],
    "wait": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
             "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAVES"],
    # share of a vector instruction's 64 lanes that the execution mask leaves on: thread-cycles over 64 x wave-cycles
    "lanes": ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VALU", "SQ_INSTS_VALU"],
    "l2": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
}
# kernels of the one-time passes and of the counted (untimed) frame are left out of the per-frame sums
NOT_FRAME = ("emit_kernel", "bound_kernel", "sort_", "scan_", "add_offsets", "karras", "node_box", "leaf_box", "root_box",
             "morton", "pack_kernel", "box_union", "bounds_kernel", "prepare_", "beam_axis", "at::native", "__amd_rocclr",
             "eval_bins_kernel<true,", "eval_kernel<true,", "ray_setup_kernel<true>", "lbvh_", "small_sort", "ncclDevKernel")


def source_stamp():
    sys.path.insert(0, ROOT)
    from raytracer_amd.api import source_stamp as stamp
    return stamp()


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def is_frame_kernel(name):
    return not any(s in name for s in NOT_FRAME)


def run_pass(workload, out_dir, name, extra):
    d = os.path.join(out_dir, f"{workload}_{name}")
    cmd = ["rocprofv3"] + extra + ["-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload,
                                   "--only", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--in-flight", "0"]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    open(os.path.join(out_dir, f"{workload}_{name}.log"), "w").write(r.stdout)
    if r.returncode != 0:
        raise SystemExit(f"pass {name} failed ({r.returncode}); see {out_dir}/{workload}_{name}.log")
    return d


def kernel_stats(d):
    tot = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for path in glob.glob(d + "/**/*.db", recursive=True):
        con = sqlite3.connect(path)
        for name, dur in con.execute("select name, duration from kernels"):
            t = tot[short(name)]
            t[0] += 1; t[1] += dur; t[2] = min(t[2], dur); t[3] = max(t[3], dur)
    total = sum(t[1] for t in tot.values())
    rows = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for k, t in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        rows.append(f'"{k}",{t[0]},{int(t[1])},{t[1] / t[0]:.1f},{100 * t[1] / total:.2f},{int(t[2])},{int(t[3])}')
    return "\n".join(rows) + "\n"


def counters(d):
    tot, calls = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for path in glob.glob(d + "/**/*.db", recursive=True):
        con = sqlite3.connect(path)
        for name, disp, cname, val in con.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            k = short(name)
            tot[k][cname] += float(val)
            calls[k].add(disp)
    return {k: dict({c: v / len(calls[k]) for c, v in tot[k].items()}, launches=len(calls[k])) for k in tot}


def main():
    workload, out_dir = sys.argv[1], os.path.abspath(sys.argv[2])
    passes = sys.argv[3:] or ["trace", "fp32", "lanes", "wait", "fetch", "write"]
    os.makedirs(out_dir, exist_ok=True)
    per_kernel = defaultdict(dict)
    for name in passes:
        if name == "trace":
            d = run_pass(workload, out_dir, name, ["--kernel-trace"])
            open(os.path.join(out_dir, f"{workload}_kernel_stats.csv"), "w").write(kernel_stats(d))
            shutil.rmtree(d, ignore_errors=True)      # the raw databases are tens of MB per pass
            print(f"[profile_frame] {workload} trace done", flush=True)
            continue
        d = run_pass(workload, out_dir, name, ["--pmc"] + PASSES[name])
        for k, c in counters(d).items():
            per_kernel[k].update(c)
        shutil.rmtree(d, ignore_errors=True)
        print(f"[profile_frame] {workload} {name} done", flush=True)
    if not per_kernel:
        return
    # a kernel of the timed frames is launched once per step (20 + 3 warm-up); what the sizing frame and the counted
    # frame launch once or twice is left out
    frame = {k: v for k, v in per_kernel.items() if is_frame_kernel(k) and v.get("launches", 0) >= 20}
    def issued(v):
        return 64.0 * (v.get("SQ_INSTS_VALU_ADD_F32", 0) + v.get("SQ_INSTS_VALU_MUL_F32", 0) +
                       2 * v.get("SQ_INSTS_VALU_FMA_F32", 0) + v.get("SQ_INSTS_VALU_TRANS_F32", 0))

    def lane_frac(v):
        # rocprofv3's own VALUUtilization: SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)
        cyc = v.get("SQ_ACTIVE_INST_VALU", 0)
        return min(1.0, v.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * cyc)) if cyc > 0 else None
    for v in frame.values():
        v["lane_active_frac"] = lane_frac(v)
    flops = sum(issued(v) for v in frame.values())
    # FP32 operations on lanes the execution mask leaves on: each kernel's issued count times its own lane share (the
    # share is measured over all its vector instructions; the FP32 ones are taken to have the kernel's mean)
    have_lanes = all(v.get("lane_active_frac") is not None for v in frame.values()) and bool(frame)
    active = sum(issued(v) * v["lane_active_frac"] for v in frame.values()) if have_lanes else None
    fetch = sum(2 * 1024.0 * v.get("FETCH_SIZE", 0) for v in frame.values())
    write = sum(1024.0 * v.get("WRITE_SIZE", 0) for v in frame.values())
    out = {"workload": workload, "cull": "bins", "source_stamp": source_stamp(),
           "method": "rocprofv3 --pmc, one pass per counter group, over `python3 bench.py --workload W --only --steps 20 --warmup 3 "
                     "--no-cpu-baseline --in-flight 0`; per kernel the mean per launch; frame sums leave out the one-time "
                     "passes and the counted frame; fp32_flop = 64 x (ADD_F32 + MUL_F32 + 2 FMA_F32 + TRANS_F32) wave-level "
                     "instructions; FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)",
           "fp32_flop_per_frame": flops if "fp32" in passes else None,
           "fp32_flop_active_lanes_per_frame": active if ("fp32" in passes and "lanes" in passes) else None,
           "lane_active_frac": (active / flops) if (active is not None and flops > 0 and "fp32" in passes) else None,
           "fetch_bytes": fetch if "fetch" in passes else None, "write_bytes": write if "write" in passes else None,
           "traffic_bytes": fetch + write if ("fetch" in passes and "write" in passes) else None,
           "frame_kernels": frame, "other_kernels": {k: v for k, v in per_kernel.items() if k not in frame}}
    json.dump(out, open(os.path.join(out_dir, f"{workload}_pmc.json"), "w"), indent=1)
    print(f"[profile_frame] wrote {out_dir}/{workload}_pmc.json: fp32 flop/frame {flops:.4g}, traffic {fetch + write:.4g} B", flush=True)


if __name__ == "__main__":
    main()
