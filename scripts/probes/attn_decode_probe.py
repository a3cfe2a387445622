"""Decode attention (shared-prefix + tail) time vs tail length at the cfg3 shape: 8 prompts x 8 rollouts, P = 1402, 28/4 heads.
Buffers of 28 layers are rotated so every call sees cold KV like the real decode loop."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
nP, Kn, P, C, Hq, Hkv, D, L = 8, 8, 1402, 512, 28, 4, 128, 28
B = nP * Kn
pk = [torch.randn(nP, P, Hkv, D, device=dev).bfloat16() for _ in range(L)]
pv = [torch.randn(nP, P, Hkv, D, device=dev).bfloat16() for _ in range(L)]
tk = torch.randn(L, B, C, Hkv, D, device=dev).bfloat16()
tv = torch.randn(L, B, C, Hkv, D, device=dev).bfloat16()
q = torch.randn(B, Hq * D, device=dev).bfloat16()
plen = torch.full((nP,), P, dtype=torch.int32, device=dev)
prompt_of = (torch.arange(B, device=dev) // Kn).int()
ws = torch.empty(K.attn_decode_workspace_bytes(nP, Hkv) // 4, device=dev)
o = torch.empty(B, Hq * D, device=dev, dtype=torch.bfloat16)
for tl in (0, 63, 127, 255, 383, 511):
    tail_len = torch.tensor([tl], dtype=torch.int32, device=dev)
    def run():
        for i in range(L):
            K.attn_decode_shared(q, pk[i], pv[i], plen, prompt_of, tk[i], tv[i], tail_len, Kn, Hq, Hkv, D, D ** -0.5, out=o, workspace=ws)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"tail_len {tl:4d}: {e0.elapsed_time(e1) / (10 * L) * 1e3:6.1f} us per layer (prefix + tail kernels)")
