"""GPU: the BASELINE.json configurations that are not the headline, as far as ONE GPU allows (configs[3] and [4] are DP = 8 jobs:
what a single rank of them runs is covered here; the exchange itself by the gloo world-2 tests and tests/test_multigpu_gpu.py).

  * cfg5 (configs[4], Qwen2-VL-7B, 32 frames @ 448^2 -> grid (16, 32, 32), P = 4458): the LLM side -- shared-prefix scoring of two
    rollouts equals the rollouts scored alone (bit for bit with the GEMM's K-split tail off), ONE FULL STEP (rollout -> reference /
    policy scoring -> loss -> backward -> AdamW) finite with the weights moved, peak memory asserted;
  * cfg4 (configs[3]: ONE prompt group per GPU, an 8-row decode batch -- the reference's own launch shape, SC:21,39): hipGraph
    replay == eager launches, token for token; one full step;
  * cfg2 (configs[1]: Qwen2-VL-2B, 8 frames, K = 4, C = 512, 4 groups): shared prefix == rows at the real shapes, greedy decode of
    the K copies coincides, two groups per scoring pass == group by group.
No oracle at these sizes in seconds: size-independent properties; the oracle comparisons at these WIDTHS are
tests/test_layer_local_gpu.py, at full 2B depth tests/test_depth_gpu.py / test_precise_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from spacer_amd import kernels as K                                         # noqa: E402
from spacer_amd.grpo import GRPOEngine, GRPOHyper, group_advantages          # noqa: E402
from spacer_amd.qwen2vl.config import QWEN2_VL_2B, QWEN2_VL_7B               # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine                          # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, random_init_              # noqa: E402
from spacer_amd.rollout import RolloutEngine, SamplingParams                 # noqa: E402
from spacer_amd.synthetic import make_prompt, synthetic_rewards              # noqa: E402


@pytest.fixture(scope="module")
def ge7(dev):
    """The whole training state of a 7B replica: policy + frozen reference + fp32 master + Adam moments + gradients (165 GB)."""
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    params = FlatParams.empty(QWEN2_VL_7B, dev)
    random_init_(params, seed=1234)
    ge = GRPOEngine(QWEN2_VL_7B, params, GRPOHyper(num_generations=8, temporal=False))
    yield ge
    del ge, params
    torch.cuda.empty_cache()


def _full_step(ge, prompts, C, step_idx=0):
    dev = ge.dev
    Kn = ge.h.num_generations
    before = ge.policy.flat[:4096].clone()
    sp = SamplingParams(max_new_tokens=C, seed=5, suppress_eos=True)
    comp = ge.rollout(prompts, sp)
    assert tuple(comp.shape) == (len(prompts) * Kn, C)
    losses = []
    for g, pr in enumerate(prompts):
        adv, _ = group_advantages(synthetic_rewards(step_idx, g, Kn).sum(1), Kn)
        res = ge.score_and_backward(pr, comp[g * Kn:(g + 1) * Kn], adv.to(dev), grad_scale=1.0 / len(prompts))
        losses.append(res["loss"])
        assert torch.isfinite(res["logps"]).all() and torch.isfinite(res["ref_logps"]).all()
        assert torch.equal(res["logps"], res["ref_logps"]) or step_idx > 0        # step 0: policy == reference, same kernels
    ge.optimizer_step()
    norm = ge.grad_norm()
    assert all(torch.isfinite(l).all() for l in losses) and norm > 0 and norm == norm
    assert not torch.equal(before, ge.policy.flat[:4096])
    return comp


def test_cfg5_llm_side_shared_prefix_and_one_full_step(ge7, dev, monkeypatch):
    cfg, eng = ge7.cfg, ge7.engine
    prompt, _ = make_prompt(cfg, 0, 32, 448, 448, 360, dev)
    assert tuple(prompt.grids[0]) == (16, 32, 32) and prompt.ids.numel() == 4096 + 362
    comps = torch.randint(1000, 150000, (2, 64), generator=torch.Generator().manual_seed(3)).to(dev)
    with K.plan(gemm_no_split=1):
        lp = eng.score_group(prompt.ids, comps, prompt.pix, prompt.grids)
        for k in range(2):
            alone = eng.score_group(prompt.ids, comps[k:k + 1], prompt.pix, prompt.grids)
            assert torch.equal(lp[k], alone[0]), float((lp[k] - alone[0]).abs().max())
    assert torch.isfinite(lp).all()
    _full_step(ge7, [prompt], 24)
    peak = torch.cuda.max_memory_allocated() / 1e9
    print(f"cfg5 single group (P = 4458, K = 8, C = 24): peak {peak:.1f} GB")
    assert peak < 260.0           # 165 GB of training state + one 4.7k-token group's activations + the ViT over 16 384 patches


def test_cfg4_shape_graph_equals_eager_and_one_full_step(ge7, dev, monkeypatch):
    cfg = ge7.cfg
    prompt, _ = make_prompt(cfg, 1, 16, 280, 364, 360, dev)
    with K.plan(skinny_blocks=1):      # decode GEMMs without split-K atomics: reproducible rollouts
        ge7.roll.invalidate()
        sp = SamplingParams(max_new_tokens=12, seed=9, suppress_eos=True)
        a = ge7.roll.generate([prompt], 8, sp, use_graph=True)           # ONE prompt group: an 8-row decode batch
        b = ge7.roll.generate([prompt], 8, sp, use_graph=False)
    assert tuple(a.shape) == (8, 12) and torch.equal(a, b)
    assert len({tuple(r.tolist()) for r in a}) > 1                    # sampled rollouts differ from one another
    _full_step(ge7, [prompt], 16, step_idx=1)


def test_cfg2_workload_shapes(dev, monkeypatch):
    cfg = QWEN2_VL_2B
    params = FlatParams.empty(cfg, dev)
    random_init_(params, seed=1234)
    eng = Qwen2VLEngine(cfg, params)
    prompts = [make_prompt(cfg, g, 8, 280, 364, 360, dev)[0] for g in range(2)]
    assert tuple(prompts[0].grids[0]) == (4, 20, 26) and prompts[0].ids.numel() == 520 + 362
    comps = [torch.randint(1000, 150000, (4, 512), generator=torch.Generator().manual_seed(3 + g)).to(dev) for g in range(2)]
    with K.plan(gemm_no_split=1):
        lp = eng.score_group(prompts[0].ids, comps[0], prompts[0].pix, prompts[0].grids)
        alone = eng.score_group(prompts[0].ids, comps[0][2:3], prompts[0].pix, prompts[0].grids)
        assert torch.equal(lp[2], alone[0])
        both = eng.score_groups([(p.ids, p.pix, p.grids) for p in prompts], comps)      # two groups per pass == group by group
        lp1 = eng.score_group(prompts[1].ids, comps[1], prompts[1].pix, prompts[1].grids)
        assert torch.equal(both[:4], lp) and torch.equal(both[4:], lp1)
    roll = RolloutEngine(eng)
    with K.plan(skinny_blocks=1):
        out = roll.generate([prompts[0]], 4, SamplingParams(max_new_tokens=8, top_k=1, top_p=1.0, suppress_eos=True), use_graph=True)
    assert tuple(out.shape) == (4, 8) and all(torch.equal(out[0], out[k]) for k in range(1, 4))
    sc = eng.score_group(prompts[0].ids, out, prompts[0].pix, prompts[0].grids)
    alts = out[:1].repeat(6, 1)
    alts[1:, -1] = torch.randint(1000, 150000, (5,), generator=torch.Generator().manual_seed(4)).to(dev)
    sa = eng.score_group(prompts[0].ids, alts, prompts[0].pix, prompts[0].grids)
    assert float(sa[0, -1]) >= float(sa[1:, -1].max()) - 3e-2 and torch.isfinite(sc).all()
    del roll, eng, params
    torch.cuda.empty_cache()
