"""What do the attention kernels' empty workgroups cost?  The grids are num_segs x ceil(max_q / tile) [x num_segs for dK/dV]: with a 1402-row
prompt and 512-row completions most (segment, block, attending segment) triples have nothing to do and exit at once.  Probe: the cfg3
two-group layout (18 segments) against the same layout with 16 more one-row segments appended (34 segments: the real work grows by 16 rows,
the launched workgroups by 73 k for dK/dV, 9.9 k for dQ, 4.9 k for the forward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
P, C, Kn, Hq, Hkv, D = 1402, 512, 8, 28, 4, 128
one = [(0, P, 0, 0)] + [(P + i * C, C, 0, P) for i in range(Kn)]
two = one + [(5498 + s[0], s[1], 5498 + s[2] if s[3] else 0, s[3]) for s in one]
T0 = 2 * 5498
extra = [(T0 + i, 1, 0, 0) for i in range(16)]
qd, kd = Hq * D, Hkv * D

def run(seglist, T):
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev).bfloat16()
    q, k, v = qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:]
    segs = K.make_segments(seglist, dev)
    d_o = torch.randn(T, qd, device=dev).bfloat16()
    res = {}
    for rep in range(3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        for _ in range(5):
            o, lse = K.attn_fwd(q, k, v, segs, P, Hq, Hkv, D, True, D ** -0.5)
        ev[1].record()
        for _ in range(5):
            dq = torch.empty_like(qkv)[:, :qd]
            dk, dv = torch.zeros(T, kd, device=dev), torch.zeros(T, kd, device=dev)
            K.attn_bwd(q, k, v, o, d_o, lse, segs, P, Hq, Hkv, D, True, D ** -0.5, dq=dq, dk32=dk, dv32=dv)
        ev[2].record(); torch.cuda.synchronize()
        res = {"fwd": min(res.get("fwd", 1e9), ev[0].elapsed_time(ev[1]) / 5 * 1e3), "bwd": min(res.get("bwd", 1e9), ev[1].elapsed_time(ev[2]) / 5 * 1e3)}
    return res

a = run(two, T0)
b = run(two + extra, T0 + 16)
print(f"18 segments: forward {a['fwd']:.0f} us, backward (delta + dQ + dK/dV + 2 zero fills) {a['bwd']:.0f} us")
print(f"34 segments: forward {b['fwd']:.0f} us, backward {b['bwd']:.0f} us   (+16 rows of real work; +4.9 k / +9.9 k / +73 k empty workgroups)")

# ---- the forward as TWO launches: completion segments (max_q = 512, no empty query blocks) then prompt segments (max_q = 1402)
T = T0
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev).bfloat16()
q, k, v = qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:]
comp = K.make_segments([s for s in two if s[3] > 0], dev)
prom = K.make_segments([s for s in two if s[3] == 0], dev)
allsegs = K.make_segments(two, dev)
o = torch.empty(T, qd, device=dev, dtype=torch.bfloat16); lse = torch.empty(Hq, T, device=dev)
o2 = torch.empty_like(o); lse2 = torch.empty_like(lse)
best = [1e9, 1e9]
for rep in range(4):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(5):
        K.attn_fwd(q, k, v, allsegs, P, Hq, Hkv, D, True, D ** -0.5, out=o, lse=lse)
    ev[1].record()
    for _ in range(5):
        K.attn_fwd(q, k, v, comp, C, Hq, Hkv, D, True, D ** -0.5, out=o2, lse=lse2)
        K.attn_fwd(q, k, v, prom, P, Hq, Hkv, D, True, D ** -0.5, out=o2, lse=lse2)
    ev[2].record(); torch.cuda.synchronize()
    best = [min(best[0], ev[0].elapsed_time(ev[1]) / 5 * 1e3), min(best[1], ev[1].elapsed_time(ev[2]) / 5 * 1e3)]
print(f"forward, one launch over 18 segments: {best[0]:.0f} us; completions then prompts as two launches: {best[1]:.0f} us; same bits: {bool(torch.equal(o, o2))}")
