import os, sys, time, torch
sys.path.insert(0, '.')
from oracle import qwen2vl_fp32 as O
one = O.make_config(hidden=3584, layers=1, heads=28, kv_heads=4, intermediate=18944, vocab=1024, vit_dim=1280, vit_depth=1, vit_heads=16, vit_mlp=5120, head_dim=128)
w = O.random_weights(one, seed=1)
for T in (256, 1024):
    x = torch.randn(T, 3584) * 0.02; pos = torch.arange(T).view(1, T).expand(3, T)
    fl = 2 * T * (3584 * 4608 + 3584 * 3584 + 3 * 3584 * 18944) + 4 * T * T * 28 * 128 // 2
    for th in (16, 32, 64, 128, 256):
        torch.set_num_threads(th)
        with torch.no_grad():
            O.llm_forward(w, one, x, pos, return_hidden=True)
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 3: O.llm_forward(w, one, x, pos, return_hidden=True); n += 1
            dt = time.perf_counter() - t0
        print(T, th, f"{fl * n / dt / 1e9:.1f} GFLOP/s")
