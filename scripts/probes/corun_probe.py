#!/usr/bin/env python
"""can the reference model's prompt-side forward (ViT + LLM over the prompt tokens: MFMA-bound, depends on the
prompt only) hide under the HBM-bound decode loop on a second stream?  Measures, at the cfg3 shapes (Qwen2-VL-7B, 8 prompts, 64 decode
rows), the decode ms per token-step and the wall time of [decode loop + reference prompt forward]:
  serial        decode loop, then the reference prompt forward on the same stream (today's order of work)
  side          the reference prompt forward on a second, low-priority stream while the decode graph replays
  side-mask N   the same with the side stream restricted to N compute units (hipExtStreamCreateWithCUMask) and the decode GEMMs
                planned for 256 - N (spacer_plan::cus)
Prints one JSON line per mode.       python scripts/probes/corun_probe.py [C=128]"""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spacer_amd import kernels as K                                    # noqa: E402
from spacer_amd.qwen2vl.config import PRESETS                          # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine                    # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, random_init_       # noqa: E402
from spacer_amd.rollout import RolloutEngine, SamplingParams          # noqa: E402
from spacer_amd.synthetic import make_prompt                          # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
cfg = PRESETS["Qwen2-VL-7B"]
params = FlatParams.empty(cfg, dev)
random_init_(params, seed=1234)
ref = FlatParams(cfg, params.flat.clone(), params.specs)
eng, ref_eng = Qwen2VLEngine(cfg, params), Qwen2VLEngine(cfg, ref)
roll = RolloutEngine(eng)
prompts = [make_prompt(cfg, g, 16, 280, 364, 360, dev)[0] for g in range(8)]
sp = SamplingParams(max_new_tokens=C, seed=3, suppress_eos=True)
side_roll = RolloutEngine(ref_eng)          # its _prefill = ViT + prompt forward of all prompts in one packed pass (+ prompt K/V)


def ref_prompt_forward():
    with torch.no_grad():
        side_roll._prefill(prompts, False)


def masked_stream(n_cus: int, stride: int):
    """A HIP stream whose kernels may only run on n_cus compute units: bits [0, n_cus) x stride of the CU mask."""
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * 8)()
    for i in range(n_cus):
        b = i * stride
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(mode: str, side=None, cus: int = 0):
    import threading
    K.PLAN.cus = cus
    roll.invalidate()
    stats = {}
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    started = torch.cuda.Event()
    worker = []

    def side_work():                                        # a second host thread: the main one is busy replaying the decode graph
        torch.cuda.set_device(dev)                          # (hipGraphLaunch costs about as much host time as the step takes on the GPU)
        with torch.cuda.stream(side):
            side.wait_event(started)
            s0.record()
            ref_prompt_forward()
            s1.record()

    def mark():                                             # right after the decode graph's capture (which synchronises the device)
        started.record()
        if side is not None:
            worker.append(threading.Thread(target=side_work))
            worker[0].start()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    roll.generate(prompts, 8, sp, stats=stats, on_decode_start=mark)
    t_enq = time.perf_counter() - t0
    if side is None:
        s0.record()
        ref_prompt_forward()
        s1.record()
    else:
        worker[0].join()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    a, b, c = stats["events"][0]
    out = {"mode": mode, "C": C, "wall_ms": round(1e3 * wall, 1), "policy_prefill_ms": round(a.elapsed_time(b), 1),
           "decode_loop_ms": round(b.elapsed_time(c), 1), "decode_ms_per_token_step": round(b.elapsed_time(c) / max(1, stats["decode_steps"]), 3),
           "ref_prompt_forward_ms": round(s0.elapsed_time(s1), 1), "decode_plan_cus": cus or 256,
           "host_generate_ms": round(1e3 * t_enq, 1)}
    print(json.dumps(out), flush=True)
    K.PLAN.cus = 0


run("warm-up (serial)")
run("serial")
lo = torch.cuda.Stream(device=dev, priority=0)
run("side stream, no mask", side=lo)
for n, stride, cus in ((32, 1, 224), (32, 1, 0), (64, 1, 192), (64, 1, 0), (16, 1, 240), (16, 1, 0), (128, 1, 0)):
    try:
        run(f"side stream masked to {n} CUs (mask bit stride {stride}), decode planned for {cus or 256} CUs", side=masked_stream(n, stride), cus=cus)
    except Exception as exc:      # noqa: BLE001
        print(json.dumps({"mode": f"mask {n}/{stride}", "error": f"{type(exc).__name__}: {exc}"[:200]}), flush=True)
run("serial (again)")
