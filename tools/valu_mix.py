#!/usr/bin/env python3
"""VALU issue cycles of one (wave, level) update of the streaming kernel, from the compiler's assembly.

    python tools/valu_mix.py            -> profiles/valu_mix.json

bench.py's VALU line needs, per kernel variant, the issue cycles one wave spends on the vector ALU per update of its row
(64 lanes x V cells, one time level).  The level loop of k_stream is one basic-block chain that the compiler does not unroll
(S is a run-time argument), so its instructions can be counted directly: this script compiles the two instantiation units of
the kernel to assembly (hipcc -S --cuda-device-only, same flags as the library), finds in every kernel the innermost loop that
reads the neighbour rows from LDS (ds_read_b128 / ds_read_b64), counts its instructions by class and prices each class with the
issue costs measured by tools/valu_issue.hip on an MI355X (cycles per wave-instruction and SIMD at 2 - 4 waves per SIMD):

    packed fp32 (v_pk_*_f32)            4        fp64 add / mul / fma                 4
    other fp32 / integer / move         2        v_rcp_f32, v_sqrt_f32, v_rsq_f32      8
    DPP move (wave_shr / wave_shl)      4        v_rcp_f64, v_rsq_f64, v_sqrt_f64     16

LDS instructions and scalar instructions are counted but not priced (they issue on other ports).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "latticeboltzmannsimulations_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only"]
COLL = {0: ("SRT", "strict"), 1: ("TRT", "strict"), 2: ("MRT", "strict"), 3: ("MRT", "fast"), 4: ("SRT", "fast"), 5: ("TRT", "fast")}


def classify(op):
    if op.startswith("v_pk_") and "f32" in op:
        return "pk_f32", 4
    if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
        return "trans_f64", 16
    if re.match(r"v_(rcp|rsq|sqrt|exp|log)_f32", op):
        return "trans_f32", 8
    if re.match(r"v_(add|mul|fma|min|max)_f64", op) or op.startswith("v_div_") and "f64" in op:
        return "f64", 4
    if op.endswith("_dpp"):
        return "dpp", 4
    if op.startswith("v_"):
        return "valu", 2
    if op.startswith("ds_"):
        return "lds", 0
    if op.startswith("s_"):
        return "salu", 0
    if op.startswith(("global_", "scratch_", "buffer_")):
        return "vmem", 0
    return "other", 0


def level_loop(lines):
    """(start, end) line indices of the level loop: the shortest loop (backward branch) that reads the neighbour rows from LDS,
    shifts lanes by DPP, contains the workgroup barrier and touches no global memory (the frame passes, which share the kernel,
    have loops with LDS reads too)."""
    labels = {m.group(1): i for i, ln in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", ln))}
    best = None
    for i, ln in enumerate(lines):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if not m or m.group(1) not in labels or labels[m.group(1)] >= i:
            continue
        a, b = labels[m.group(1)], i
        body = lines[a:b + 1]
        if not any(re.match(r"\s+ds_read_b(128|64)", x) for x in body):
            continue
        if not any("_dpp" in x for x in body) or not any(re.match(r"\s+s_barrier", x) for x in body):
            continue
        if any(re.match(r"\s+(global_|scratch_|buffer_)", x) for x in body):
            continue
        if best is None or b - a < best[1] - best[0]:
            best = (a, b)
    return best


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for unit, dtype in (("lbm_stream_f32.hip", "float32"), ("lbm_stream_f64.hip", "float64"),
                            ("lbm_streamw_f32.hip", "float32"), ("lbm_streamw_f64.hip", "float64")):   # k_stream, k_stream_walls (r03)
            asm = os.path.join(tmp, unit + ".s")
            subprocess.check_call([HIPCC] + FLAGS + [os.path.join(CSRC, unit), "-o", asm], stderr=subprocess.DEVNULL)
            text = open(asm).read().split("\n")
            starts = [(i, m.group(1)) for i, ln in enumerate(text) if (m := re.match(r"^(_Z\d+k_stream\w+):", ln))]
            for n, (i, sym) in enumerate(starts):
                end = starts[n + 1][0] if n + 1 < len(starts) else len(text)
                m = re.match(r"_Z8k_streamI([fd])Li(\d)ELi(\d)ELb([01])E", sym)
                mw = re.match(r"_Z14k_stream_wallsI([fd])Li(\d)ELb([01])E", sym)
                if not m and not mw:
                    continue
                kname = "k_stream" if m else "k_stream_walls"
                coll, sem, turb = (int(m.group(2)), int(m.group(3)), int(m.group(4))) if m else (int(mw.group(2)), 1, int(mw.group(3)))
                if sem != 1:
                    continue                      # (MRT.py semantics shares the tile code; report MRT_GPU)
                body = text[i:end]
                loop = level_loop(body)
                if loop is None:
                    continue
                counts, cycles = {}, 0
                for ln in body[loop[0]:loop[1] + 1]:
                    mm = re.match(r"\s+([a-z_0-9]+)", ln)
                    if not mm or ln.strip().startswith((";", ".")):
                        continue
                    cls, c = classify(mm.group(1))
                    counts[cls] = counts.get(cls, 0) + 1
                    cycles += c
                rt, arith = COLL[coll]
                key = f"{kname}:{dtype}:{rt}:{arith}:turb{turb}"
                out[key] = {"issue_cycles_per_wave_update": cycles, "instructions": counts,
                            "source": "tools/valu_mix.py: level loop of " + sym + ", issue costs of tools/valu_issue.hip (profiles/r02_logs/valu_issue.log)"}
    path = os.path.join(ROOT, "profiles", "valu_mix.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items()):
        print(f"{k:44s} {v['issue_cycles_per_wave_update']:5d} cycles  {v['instructions']}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
