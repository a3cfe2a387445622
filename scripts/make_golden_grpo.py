"""Generate tests/golden/grpo_lines.json by EXECUTING the reference's own text for the GRPO lines of ``SGRLVRTrainer.compute_loss``
(SG_RLVR_trainer.py:357-366 + 528 per-token log-probs of a stub model's logits and the completion slice, :493-498 first-EOS mask, :551-552 k3 KL, :598-643 T-GRPO bonus / length bonus / group advantage / loss, :650-683 the
logged metrics of an emulated multi-rank world) on seeded
cases.  The module cannot be imported here (top-level ``import trl`` / ``qwen_vl_utils``), so the three line ranges are cut out of the
file (anchors asserted, so that drift of the reference is loud), dedented and exec'd against a stub ``self`` and CPU tensors;
``.to('cuda')`` (TR:610-613) is redirected to the CPU for the duration of the exec.  Data only: inputs and the reference's outputs
(plus d loss / d per_token_logps from autograd THROUGH the reference's own expression).

    python scripts/make_golden_grpo.py          # needs /root/reference (authoring container only)
"""
import json
import os
import textwrap
import types

import numpy as np
import torch

REF = "/root/reference/SpaceR-SG-RLVR/src/r1-v/src/open_r1/trainer/SG_RLVR_trainer.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "grpo_lines.json")

# (first line, last line, {line: text the line must contain})  -- 1-indexed, inclusive
RANGES = {
    "mask": (493, 498, {493: "is_eos = completion_ids == self.processing_class.eos_token_id", 498: "completion_mask = (sequence_indices <= eos_idx.unsqueeze(1)).int()"}),
    "kl": (551, 552, {551: "x_clamped = torch.clamp(ref_per_token_logps - per_token_logps, min=-10, max=10)", 552: "per_token_kl = torch.exp(x_clamped) - x_clamped - 1"}),
    "logps": (357, 366, {357: "logits = model(input_ids, **kwargs).logits", 358: "logits = logits[:, :-1, :]", 363: "log_probs = logits_row.log_softmax(dim=-1)",
                         366: "return torch.stack(per_token_logps)"}),
    "slice": (528, 528, {528: "per_token_logps = per_token_logps[:, prompt_length - 1 :]"}),
    "metrics": (650, 683, {650: "completion_length = self.accelerator.gather_for_metrics(completion_mask.sum(1)).float().mean().item()",
                           665: "wrong_devices = (rewards_per_device <= 1).all(dim=1)", 683: 'self._metrics["kl"].append(self.accelerator.gather_for_metrics(mean_kl).mean().item())'}),
    "loss": (598, 643, {598: "if self.temporal and video_inputs:", 620: "if self.len_control:", 638: "advantages = (rewards - mean_grouped_rewards) / (std_grouped_rewards + 1e-4)",
                        643: "loss = ((per_token_loss * completion_mask).sum(dim=1) / completion_mask.sum(dim=1)).mean()"}),
}


def cut(lines, name):
    a, b, anchors = RANGES[name]
    for ln, text in anchors.items():
        assert text in lines[ln - 1], f"reference drifted: line {ln} is {lines[ln - 1]!r}, expected to contain {text!r}"
    return compile(textwrap.dedent("\n".join(lines[a - 1:b])), f"{REF}:{a}-{b}", "exec")


def f32(t):
    """Nested lists of the SHORTEST decimals that round-trip the float32 values."""
    a = np.asarray(t.detach().cpu().numpy(), dtype=np.float32)
    return np.vectorize(lambda x: float(str(x)), otypes=[object])(a).tolist() if a.ndim else float(str(a))


class _CpuCuda:
    """Tensor.to('cuda') -> CPU while the reference's lines run (TR:610-613 create a scalar on 'cuda')."""

    def __enter__(self):
        self.orig = torch.Tensor.to

        def to(t, *args, **kw):
            args = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in args)
            return self.orig(t, *args, **kw)
        torch.Tensor.to = to

    def __exit__(self, *exc):
        torch.Tensor.to = self.orig


def run_mask(code, ids, eos):
    me = types.SimpleNamespace(processing_class=types.SimpleNamespace(eos_token_id=eos), accelerator=types.SimpleNamespace(device="cpu"))
    ns = {"torch": torch, "self": me, "completion_ids": ids}
    exec(code, ns)
    return ns["completion_mask"]


def run_loss(code_kl, code_loss, *, temporal, video, len_control, K, beta, rpf, srpf, mask, lp, ref):
    me = types.SimpleNamespace(temporal=temporal, len_control=len_control, num_generations=K, beta=beta)
    lp = lp.clone().requires_grad_(True)
    ns = {"torch": torch, "self": me, "video_inputs": [object()] if video else None, "rewards_per_func": rpf.clone(),
          "shuffled_rewards_per_func": None if srpf is None else srpf.clone(), "completion_mask": mask, "per_token_logps": lp,
          "ref_per_token_logps": ref}
    with _CpuCuda():
        exec(code_kl, ns)
        exec(code_loss, ns)
    ns["loss"].backward()
    return dict(per_token_kl=f32(ns["per_token_kl"]) if lp.numel() <= 512 else None, temporal_rewards=f32(ns["temporal_rewards"].reshape(())), rewards=f32(ns["rewards"]),
                advantages=f32(ns["advantages"]), loss=f32(ns["loss"]), dlogp=f32(lp.grad))


def run_logps(lines, logits, input_ids, prompt_length):
    """TR:357-366 (the body of ``_get_per_token_logps``, wrapped back into a function because it ends in ``return``) on a stub model that
    returns the given logits, then TR:528 (the completion slice)."""
    a, b, anchors = RANGES["logps"]
    for ln, text in anchors.items():
        assert text in lines[ln - 1], f"reference drifted: line {ln} is {lines[ln - 1]!r}"
    body = textwrap.dedent("\n".join(lines[a - 1:b]))
    src = "def _get_per_token_logps(model, input_ids, **kwargs):\n" + textwrap.indent(body, "    ")
    ns = {"torch": torch}
    exec(compile(src, f"{REF}:{a}-{b}", "exec"), ns)
    model = lambda ids, **kw: types.SimpleNamespace(logits=logits)  # noqa: E731
    per_token_logps = ns["_get_per_token_logps"](model, input_ids)
    ns2 = {"per_token_logps": per_token_logps, "prompt_length": prompt_length}
    exec(cut(lines, "slice"), ns2)
    return per_token_logps, ns2["per_token_logps"]


def run_metrics(code, ranks, *, temporal, K, func_names):
    """TR:650-683 on every rank of an emulated world: ``accelerator.gather_for_metrics`` returns the concatenation over ranks of the
    k-th gathered tensor (pass 1 records what every rank hands to its k-th gather call -- the control flow is rank-independent --, pass 2
    replays with the concatenations; 0-dim tensors gather to one element per rank, as accelerate does).  Returns rank 0's metric dict
    (asserted equal on every rank)."""
    import collections

    def funcs():
        out = []
        for n in func_names:
            f = lambda **kw: None  # noqa: E731
            f.__name__ = n
            out.append(f)
        return out

    def one(rank, gather):
        r = ranks[rank]
        me = types.SimpleNamespace(accelerator=types.SimpleNamespace(gather_for_metrics=gather), _metrics=collections.defaultdict(list),
                                   reward_funcs=funcs(), num_generations=K, temporal=temporal)
        ns = {"torch": torch, "self": me, "PreTrainedModel": type("PreTrainedModel", (), {}), "completion_mask": r["mask"],
              "rewards_per_func": r["rpf"], "rewards": r["rewards"], "temporal_rewards": r["temporal_rewards"],
              "std_grouped_rewards": r["std"], "per_token_kl": r["kl"]}
        exec(code, ns)
        return {k: v[0] for k, v in me._metrics.items()}
    record = [[] for _ in ranks]
    for rk in range(len(ranks)):
        one(rk, lambda t, rk=rk: (record[rk].append(t.clone()), t)[1])
    n_calls = len(record[0])
    assert all(len(x) == n_calls for x in record)
    outs = []
    for rk in range(len(ranks)):
        k = [0]

        def gather(t, k=k):
            parts = [record[q][k[0]] for q in range(len(ranks))]
            k[0] += 1
            return torch.cat([p.reshape(1) if p.dim() == 0 else p for p in parts])
        outs.append(one(rk, gather))
    assert all(o == outs[0] for o in outs)
    return outs[0]


def main():
    lines = open(REF, encoding="utf-8").read().split("\n")
    code_mask, code_kl, code_loss, code_metrics = cut(lines, "mask"), cut(lines, "kl"), cut(lines, "loss"), cut(lines, "metrics")
    g = torch.Generator().manual_seed(20260929)
    R = lambda *shape: torch.rand(*shape, generator=g)  # noqa: E731
    cases = {"mask": [], "step": [], "metrics": [], "logps": []}

    # ---- TR:493-498: no EOS, EOS at 0, several EOS, all EOS, EOS last, one-column matrices
    eos = 7
    for i in range(60):
        Kn, C = int(torch.randint(1, 7, (1,), generator=g)), int(torch.randint(1, 24, (1,), generator=g))
        ids = torch.randint(8, 50, (Kn, C), generator=g)
        for k in range(Kn):
            mode = int(torch.randint(0, 6, (1,), generator=g))
            if mode == 1:
                ids[k, 0] = eos
            elif mode == 2:
                ids[k, torch.randint(0, C, (min(C, 3),), generator=g)] = eos
            elif mode == 3:
                ids[k, :] = eos
            elif mode == 4:
                ids[k, C - 1] = eos
            elif mode == 5:
                ids[k, int(torch.randint(0, C, (1,), generator=g))] = eos
        cases["mask"].append(dict(eos_token_id=eos, completion_ids=ids.tolist(), completion_mask=run_mask(code_mask, ids, eos).tolist()))

    # ---- TR:551-552 + 598-643
    def logps(Kn, C, spread, edges):
        lp = -3.0 * R(Kn, C) - 0.01
        ref = lp + spread * (R(Kn, C) - 0.5)
        if edges:                                      # the clamp at +-10: exactly on it, just inside, far outside
            flat, rf = lp.view(-1), ref.view(-1)
            for j, d in enumerate((10.0, -10.0, 10.5, -10.5, 9.999, -9.999, 25.0, -25.0)):
                if j < flat.numel():
                    flat[j] = float(torch.tensor(-1.0 - j))
                    rf[j] = flat[j] + d
        return lp.float(), ref.float()

    def masks(Kn, C, lengths):      # (prefix masks, as TR:493-498 produces them: the fixture stores the lengths)
        m = torch.zeros(Kn, C, dtype=torch.int32)
        for k, n in enumerate(lengths):
            m[k, :n] = 1
        return m

    def add(tag, **kw):
        out = run_loss(code_kl, code_loss, **kw)
        cases["step"].append(dict(tag=tag, temporal=kw["temporal"], video=kw["video"], len_control=kw["len_control"], num_generations=kw["K"],
                                  beta=kw["beta"], rewards_per_func=f32(kw["rpf"]), shuffled_rewards_per_func=None if kw["srpf"] is None else f32(kw["srpf"]),
                                  completion_lengths=kw["mask"].sum(1).tolist(), C=kw["mask"].shape[1], per_token_logps=f32(kw["lp"]), ref_per_token_logps=f32(kw["ref"]), **out))

    acc_choices = torch.tensor([0.0, 0.05, 0.1, 0.1000001, 0.5, 0.9, 1.0, 1.5, 1.9292893, 2.0])

    def rewards(Kn, all_same=None):
        if all_same is not None:
            return torch.tensor([[all_same, 1.0]] * Kn)
        acc = acc_choices[torch.randint(0, len(acc_choices), (Kn,), generator=g)]
        fmt = torch.randint(0, 2, (Kn,), generator=g).float()
        return torch.stack([acc, fmt], 1)

    # small random cases: every flag combination, K in {2, 4, 8}, groups of equal rewards (std = 0), clamp edges
    n = 0
    for temporal in (False, True):
        for video in (False, True):
            for len_control in (False, True):
                for rep in range(18):
                    Kn = (2, 4, 8)[rep % 3]
                    C = int(torch.randint(2, 20, (1,), generator=g))
                    lengths = [int(torch.randint(1, C + 1, (1,), generator=g)) for _ in range(Kn)]
                    rpf = rewards(Kn, all_same=(0.0, 1.0, None)[rep % 3] if rep < 6 else None)
                    srpf = rewards(max(1, Kn // 2)) if temporal else None
                    lp, ref = logps(Kn, C, (0.0, 0.2, 2.0, 30.0)[rep % 4], edges=rep % 5 == 0)
                    add(f"small{n}", temporal=temporal, video=video, len_control=len_control, K=Kn, beta=(0.04, 0.0, 0.1)[rep % 3], rpf=rpf, srpf=srpf,
                        mask=masks(Kn, C, lengths), lp=lp, ref=ref)
                    n += 1
    # 0.8x threshold ties of the T-GRPO rule (TR:604): mean(acc) == 0.8 mean(shuffled acc) exactly, just above, just below
    for i, (acc, sacc) in enumerate((([0.8, 0.8], [1.0]), ([0.4, 0.4, 0.4, 0.4], [0.5, 0.5]), ([0.79, 0.8], [1.0]), ([0.81, 0.8], [1.0]),
                                     ([0.0, 0.0], [0.0]), ([1.0, 0.0, 0.6, 0.0], [0.5, 0.5]), ([0.1, 0.1], [0.0]), ([0.05, 0.3], [0.2]))):
        Kn = len(acc)
        rpf = torch.stack([torch.tensor(acc), torch.ones(Kn)], 1)
        srpf = torch.stack([torch.tensor(sacc), torch.zeros(len(sacc))], 1)
        lp, ref = logps(Kn, 6, 0.5, False)
        add(f"tie{i}", temporal=True, video=True, len_control=False, K=Kn, beta=0.04, rpf=rpf, srpf=srpf, mask=masks(Kn, 6, [6] * Kn), lp=lp, ref=ref)
    # the length rule (TR:620-629): 320 <= len <= 512 and MORE THAN ONE rollout with acc > 0.1 -- edges 319 / 320 / 512 / 513, exactly one correct,
    # exactly two, the acc = 0.1 boundary (not > 0.1)
    Cb = 516
    for i, (lens, acc) in enumerate((([319, 320, 512, 513], [1.0, 1.0, 1.0, 1.0]), ([320, 400, 512, 100], [1.0, 0.0, 0.0, 0.0]),
                                     ([320, 400, 512, 100], [1.0, 0.0, 0.5, 0.0]), ([400, 400], [0.1, 1.0]), ([400, 400], [0.1000001, 1.0]),
                                     ([512, 516, 1, 330], [0.2, 0.9, 1.0, 0.0]), ([516, 516], [1.0, 1.0]), ([321, 511, 320, 512], [0.0, 0.0, 0.0, 0.0]))):
        Kn = len(lens)
        rpf = torch.stack([torch.tensor(acc), torch.ones(Kn)], 1)
        lp, ref = logps(Kn, Cb, 0.3, False)
        for temporal in ((False, True) if i < 3 else (False,)):
            add(f"len{i}{'t' if temporal else ''}", temporal=temporal, video=True, len_control=True, K=Kn, beta=0.04, rpf=rpf,
                srpf=rewards(max(1, Kn // 2)) if temporal else None, mask=masks(Kn, Cb, lens), lp=lp, ref=ref)
    # two groups in one batch (rewards.view(-1, K)): the advantage is per group, the bonuses per batch as the reference computes them
    for i in range(6):
        Kn = 4
        rpf = rewards(2 * Kn)
        lp, ref = logps(2 * Kn, 9, 1.0, i == 0)
        add(f"twogroups{i}", temporal=False, video=True, len_control=bool(i % 2), K=Kn, beta=0.04, rpf=rpf, srpf=None,
            mask=masks(2 * Kn, 9, [int(torch.randint(1, 10, (1,), generator=g)) for _ in range(2 * Kn)]), lp=lp, ref=ref)

    # ---- TR:650-683: the logged metrics of a world of W ranks (one prompt group of K rollouts per rank, as the reference script runs)
    for i in range(40):
        W, Kn = (1, 2, 3, 8)[i % 4], (2, 4, 8)[i % 3]
        temporal = bool(i % 2)
        C = int(torch.randint(4, 12, (1,), generator=g))
        ranks = []
        for rk in range(W):
            kind = int(torch.randint(0, 4, (1,), generator=g))      # 0 / 1: mixed, 2: every reward <= 1 (all_wrong rank), 3: every reward >= 2 (all_correct rank)
            rpf = rewards(Kn)
            if kind == 2:
                rpf = torch.stack([torch.zeros(Kn), torch.randint(0, 2, (Kn,), generator=g).float()], 1)
            if kind == 3:
                rpf = torch.stack([1.0 + R(Kn), torch.ones(Kn)], 1)
            rw = rpf.sum(1)
            lens = [int(torch.randint(1, C + 1, (1,), generator=g)) for _ in range(Kn)]
            std = rw.view(-1, Kn).std(dim=1).repeat_interleave(Kn, dim=0)
            ranks.append(dict(rpf=rpf, rewards=rw, mask=masks(Kn, C, lens), temporal_rewards=torch.tensor([(1.0, 0.0, 0.5)[int(torch.randint(0, 3, (1,), generator=g))]]),
                              std=std, kl=R(Kn, C) * 0.1, lens=lens))
        out = run_metrics(code_metrics, ranks, temporal=temporal, K=Kn, func_names=["accuracy_reward", "format_reward"])
        cases["metrics"].append(dict(world=W, num_generations=Kn, temporal=temporal, C=C,
                                     ranks=[dict(completion_lengths=r["lens"], rewards_per_func=f32(r["rpf"]), rewards=f32(r["rewards"]),
                                                 temporal_rewards=f32(r["temporal_rewards"].reshape(())), std_grouped_rewards=f32(r["std"]), per_token_kl=f32(r["kl"])) for r in ranks],
                                     metrics={k: float(v) for k, v in out.items()}))

    # ---- TR:357-366 + 528: log_softmax(logits[:, :-1])[ids[:, 1:]] and the completion slice (logits of widely different scales, a row
    # whose target carries almost all / almost none of the mass)
    for i in range(24):
        B, L, V = int(torch.randint(1, 5, (1,), generator=g)), int(torch.randint(3, 14, (1,), generator=g)), (17, 40, 129)[i % 3]
        P = int(torch.randint(1, L, (1,), generator=g))
        logits = (R(B, L, V) - 0.5) * (1.0, 8.0, 40.0, 0.01)[i % 4]
        ids = torch.randint(0, V, (B, L), generator=g)
        if i % 6 == 0:
            logits[0, 0, ids[0, 1]] += 60.0             # probability ~ 1: log-prob ~ -0
        if i % 6 == 3:
            logits[0, 0, ids[0, 1]] -= 60.0             # probability ~ e-60
        full, sliced = run_logps(lines, logits, ids, P)
        cases["logps"].append(dict(logits=f32(logits), input_ids=ids.tolist(), prompt_length=P, per_token_logps=f32(full), completion_logps=f32(sliced)))

    meta = {"source": "SpaceR-SG-RLVR/src/r1-v/src/open_r1/trainer/SG_RLVR_trainer.py lines 357-366, 493-498, 528, 551-552, 598-643, 650-683, executed by scripts/make_golden_grpo.py",
            "torch": torch.__version__, "n_mask": len(cases["mask"]), "n_step": len(cases["step"]), "n_metrics": len(cases["metrics"]), "n_logps": len(cases["logps"])}
    with open(OUT, "w") as f:
        json.dump(dict(meta=meta, **cases), f, separators=(",", ":"))
    print("wrote", OUT, meta, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
