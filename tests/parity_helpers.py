"""How the GPU tests compare captions that went THROUGH the fp16 / bf16 backbone with fp32 references.

north_star: "greedy-decode token ids bit-exact".  The decoder stage is exactly that: on identical prefixes the HIP
decoder's ids equal the oracle's, always (asserted below for every prefix of every call).  Through the backbone the
prefix itself differs from the fp32 path's by the stated ViT tolerance, and an arg-max can only follow the reference
while its top-2 logit margin exceeds what that perturbation moves the logits by.  So a caption is accepted iff

  (1) the HIP ids equal the oracle decoder's ids on the HIP path's OWN prefix, bit for bit, and
  (2) either they also equal the reference ids, or at the FIRST step t where they depart from the reference (identical
      history up to there) the departure is EXPLAINED by the prefix perturbation:
        (a) derived bound AND fixed ceilings (MARGIN_BOUND on the reference's margin, PREFIX_REL_BOUND on the prefix itself, DEPART_FRACTION
            on the count -- none of which depends on the measured error), whenever the fp32 path's prefixes are available
            (``ref_prefixes``): with L_hip / L_ref the oracle's
            logits at step t on the HIP prefix / on the fp32 prefix (same history), an arg-max can differ only if the
            reference's top-2 margin is at most 2 max_v |L_hip[v] - L_ref[v]| (each logit moved by at most that
            maximum; the two leaders can have approached each other by twice it).  The test asserts exactly that, with
            the measured perturbation -- nothing is assumed about its size;
        (b) otherwise (fixtures of the imported reference, whose prefixes were not kept): a fixed MARGIN_BOUND = 1e-3 logit
            units on the oracle's top-2 margin at that step.  Logits span about +-4 and the fixtures' median top-2 margin
            is 0.5; the largest margin at any departure observed over the ~800 captions of the suite is 1.8e-4 (round 2:
            1.8e-4 and 6.7e-5), so 1e-3 leaves a factor 5 for other boxes / seeds and is 5x tighter than round 2's 0.005.

No test accepts a fraction of wrong captions: every departure has to be explained, one by one.  The backbone-free bit-exact
statement of the whole path is tests/test_gpu_parity.py::test_e2e_fp32_backbone_mode_is_bit_exact_to_the_reference_fixture
(``vit_dtype="fp32"``: no clause at all).
"""
import json
import math
import os

import torch

MARGIN_BOUND = 1e-3
# Fixed ceilings that do NOT depend on the measured perturbation (the derived inequality alone holds for ANY prefix error,
# because step (1) already ties G to the oracle on P): a caption may depart only at a near-tie of the REFERENCE (top-2 margin <=
# MARGIN_BOUND, fixed), the HIP prefix has to sit within PREFIX_REL_BOUND of the fp32 path's (relative L2 per row; a wrong ViT /
# projection fails here however the captions fall), and at most DEPART_FRACTION of a call set's captions may depart at all.
PREFIX_REL_BOUND = {"fp16": 2e-2, "bf16": 8e-2}
DEPART_FRACTION = 0.03
# bf16 operands carry 8 significant bits where fp16 carries 11: the backbone's rounding error, the prefix error behind the T = 0.01
# projection and the logit shifts all scale by 8.  Measured on the one bf16 end-to-end set (24 captions, two builds of round 5): prefix
# error 2.0e-2 / 3.3e-2 relative L2 (fp16: 2e-3), 1 and 3 captions departing, at reference margins 2.9e-4 and 1.0e-2 with logit shifts of
# 7.5e-3 / 3.1e-2.  The ceilings leave a factor ~3 over that: margin 3e-2 (fp16: 1e-3), prefix 8e-2 (fp16: 2e-2), at most 4 of 24
# captions.  README states what this means: bf16 is a backbone mode for throughput experiments, fp16 is the default because its captions
# follow the fp32 reference (2 departures in ~790).  Used by the one bf16
# end-to-end test (test_gpu_parity.py::test_e2e_full_depth_bf16_backbone_ledger); every other test runs the fp16 default.
MARGIN_BOUND_BY = {"fp16": MARGIN_BOUND, "bf16": 3e-2}
DEPART_FRACTION_BY = {"fp16": DEPART_FRACTION, "bf16": 0.17}
REPORT = []          # one record per assert_ids_explained call; conftest.pytest_terminal_summary prints and saves them
REPORT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_parity_report.json")


def assert_ids_explained(dec_oracle, gpu_log, ref_ids, label="", ref_prefixes=None, operands="fp16"):
    """gpu_log: Patchioner.call_log entries (prefix [n, E] cuda, ids [n, 30] cuda); ref_ids: list of [n_i, 30] integer
    arrays / tensors in the same row order (the reference's or the oracle's ids through the fp32 backbone); ref_prefixes:
    the fp32 path's decoder inputs in the same order (PatchionerOracle.prefix_log), or None.
    Returns (identical captions, total)."""
    P = torch.cat([p for p, _ in gpu_log]).float().cpu()
    G = torch.cat([i for _, i in gpu_log]).cpu().long()
    R = torch.cat([torch.as_tensor(r).long().reshape(-1, G.shape[1]) for r in ref_ids])
    assert G.shape == R.shape, (label, G.shape, R.shape)
    RP = None
    if ref_prefixes is not None:
        RP = torch.cat([torch.as_tensor(p).float().reshape(-1, P.shape[1]) for p in ref_prefixes])
        assert RP.shape == P.shape, (label, RP.shape, P.shape)
    keep = torch.isfinite(P).all(dim=1)          # NaN prefixes (dummy boxes): the reference decodes garbage from NaN as well
    o_ids, _, o_margin = dec_oracle.decode_ids(P[keep], cached=True)     # same sums, keys/values kept (test_oracle_golden pins it)
    assert torch.equal(G[keep], o_ids), "%s: decoder ids differ from the oracle on identical prefixes" % label
    Gk, Rk, Pk = G[keep], R[keep], P[keep]
    RPk = RP[keep] if RP is not None else None
    worst, departed, worst_moved, prefix_rel = 0.0, 0, 0.0, None
    if RPk is not None and len(RPk):
        rel = (Pk - RPk).norm(dim=1) / RPk.norm(dim=1).clamp_min(1e-30)
        prefix_rel = float(rel.max())
        assert prefix_rel <= PREFIX_REL_BOUND[operands], ("%s: a decoder prefix is %.3e (relative L2) away from the fp32 path's, bound %.1e for "
                                                          "%s operands" % (label, prefix_rel, PREFIX_REL_BOUND[operands], operands))
    for r in (Gk != Rk).any(dim=1).nonzero().flatten().tolist():
        t = int((Gk[r] != Rk[r]).nonzero()[0])
        departed += 1
        if RPk is not None:
            hist = Rk[r, :t]
            l_hip, l_ref = dec_oracle.logits_after(Pk[r], hist), dec_oracle.logits_after(RPk[r], hist)
            top2 = l_ref.topk(2).values
            m, moved = float(top2[0] - top2[1]), float((l_hip - l_ref).abs().max())
            worst, worst_moved = max(worst, m), max(worst_moved, moved)
            assert m <= MARGIN_BOUND_BY[operands], ("%s: caption %d departs from the reference at step %d where the reference's top-2 margin "
                                                    "is %.3e (> %.1e): not a near-tie" % (label, r, t, m, MARGIN_BOUND_BY[operands]))
            assert m <= 2.0 * moved, ("%s: caption %d departs from the reference at step %d where the reference's top-2 margin is "
                                      "%.3e but the prefix perturbation moves the logits by at most %.3e: not explained"
                                      % (label, r, t, m, moved))
        else:
            m = float(o_margin[r, t])
            worst = max(worst, m)
            assert m <= MARGIN_BOUND, ("%s: caption %d departs from the reference at step %d where the top-2 margin is %.3e "
                                       "(> %.2e): not a near-tie" % (label, r, t, m, MARGIN_BOUND))
    total = int(keep.sum())
    assert departed <= max(1, math.ceil(DEPART_FRACTION_BY[operands] * total)), "%s: %d of %d captions depart from the reference" % (label, departed, total)
    REPORT.append(dict(test=os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], label=label, identical=total - departed, total=total,
                       departures=departed, worst_margin_at_departure=worst, bound="derived+fixed" if RPk is not None else "fixed 1e-3",
                       max_logit_shift_at_departure=worst_moved, max_prefix_rel_err=prefix_rel, operands=operands))
    try:
        with open(REPORT_PATH, "w") as f:
            json.dump(REPORT, f, indent=1)
    except OSError:
        pass
    print("%s: %d / %d captions identical to the reference; %d departures, largest top-2 margin at a departure %.2e (%s bound)"
          % (label, total - departed, total, departed, worst, "derived" if RPk is not None else "fixed 1e-3"))
    return total - departed, total
