"""GPU: ``spacer_completion_mask`` and ``spacer_grpo_loss`` against tests/golden/grpo_lines.json -- the outputs of the reference's OWN
lines (SG_RLVR_trainer.py:493-498, 551-552, 640-643 executed from its text by scripts/make_golden_grpo.py).  Mask and lengths
bit-exact; loss, d loss / d logp and the KL metric (TR:682, from the table's per-token KL) within 2e-6 relative plus a few fp32 ulps of
the exponential (the kernel uses the hardware exp and combines row sums with fp32 atomics)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from spacer_amd import kernels as K      # noqa: E402


def _golden():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grpo_lines.json")) as f:
        return json.load(f)


def test_completion_mask_kernel_equals_the_reference_lines(dev):
    for c in _golden()["mask"]:
        ids = torch.tensor(c["completion_ids"], dtype=torch.int64, device=dev)
        mask, lengths = K.completion_mask(ids, c["eos_token_id"])
        want = torch.tensor(c["completion_mask"], dtype=torch.int32)
        assert torch.equal(mask.cpu(), want)
        assert lengths.cpu().tolist() == want.sum(1).tolist()


def test_grpo_loss_kernel_equals_the_reference_lines(dev):
    n = 0
    for c in _golden()["step"]:
        C = c["C"]
        lens = c["completion_lengths"]
        mask = torch.zeros(len(lens), C, dtype=torch.int32)
        for k, m in enumerate(lens):
            mask[k, :m] = 1
        lp = torch.tensor(c["per_token_logps"], dtype=torch.float32, device=dev)
        ref = torch.tensor(c["ref_per_token_logps"], dtype=torch.float32, device=dev)
        adv = torch.tensor(c["advantages"], dtype=torch.float32, device=dev)
        loss, kl, dlogp = K.grpo_loss(lp, ref, adv, mask.to(dev), c["beta"])
        want_loss = c["loss"]
        # the loss is a mean of row terms of size |A| that cancel across the group (advantages sum to zero); a row term is itself a sum
        # of up to 516 fp32 values -- the bound is relative to |A| (a few fp32 ulps of the row sums, in torch's order or the kernel's)
        assert abs(float(loss) - want_loss) <= 1e-6 * abs(want_loss) + 5e-7 * max(float(adv.abs().max()), 1e-3), (c["tag"], float(loss), want_loss)
        want_g = torch.tensor(c["dlogp"], dtype=torch.float32)
        got_g = dlogp.cpu()
        scale = float(want_g.abs().max())
        # relative to the row scale: a gradient entry is (-A + beta (1 - e^x)) / (len K); e^x at the clamp edge is 22026 +- 1 ulp of the fast exp
        assert float((got_g - want_g).abs().max()) <= 2e-6 * scale + 1e-12, (c["tag"], float((got_g - want_g).abs().max()), scale)
        if c["per_token_kl"] is not None:
            pk = torch.tensor(c["per_token_kl"], dtype=torch.float32)
            want_kl = float(((pk * mask).sum(1) / mask.sum(1)).mean())                     # TR:682
            # e^x - x - 1 cancels: for |x| ~ 0.1 the value is ~ x^2 / 2 while e^x carries an fp32 rounding of ~1e-7 ABSOLUTE (in the
            # reference's torch.exp as in the kernel's hardware exp) -- so the bound has an absolute term of a few ulps of 1.0
            assert abs(float(kl) - want_kl) <= 2e-6 * abs(want_kl) + 3e-7, (c["tag"], float(kl), want_kl)
        n += 1
    assert n >= 160


def test_logprob_kernels_equal_the_reference_lines(dev):
    """``spacer_logprob_fwd`` and the chunked online form (``spacer_lse_chunk`` / ``spacer_lse_finish``: what the head runs over vocabulary
    chunks) against TR:357-366 executed on the same logits: log-probs of the completion positions within 1e-6 of the reference's fp32
    log_softmax (relative to the row's logit scale: both sides subtract a row maximum of that size)."""
    for c in _golden()["logps"]:
        logits = torch.tensor(c["logits"], dtype=torch.float32)
        ids = torch.tensor(c["input_ids"])
        B, L, V = logits.shape
        P = c["prompt_length"]
        rows = logits[:, P - 1:L - 1].reshape(-1, V)                         # the rows that predict the completion tokens
        tgt = ids[:, P:].reshape(-1).to(dev)
        want = torch.tensor(c["completion_logps"], dtype=torch.float32).reshape(-1)
        Vp = (V + 3) // 4 * 4
        buf = torch.zeros(rows.shape[0], Vp, device=dev)
        buf[:, :V] = rows.to(dev)
        lp, _ = K.logprob_fwd(buf[:, :V], tgt)
        scale = max(1.0, float(rows.abs().max()))
        assert float((lp.cpu() - want).abs().max()) <= 2e-6 * scale, (float((lp.cpu() - want).abs().max()), scale)
        state = torch.empty(3, rows.shape[0], device=dev)
        ch = 16
        for c0 in range(0, V, ch):
            c1 = min(V, c0 + ch)
            K.lse_chunk_(buf[:, c0:c1], tgt, c0, state, first=c0 == 0)
        lp2, _ = K.lse_finish(state)
        assert float((lp2.cpu() - want).abs().max()) <= 2e-6 * scale
