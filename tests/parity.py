"""Shared parity criteria (used by the oracle-vs-golden CPU tests and the HIP-vs-golden GPU tests).

BASELINE.json's tolerance is 1e-3 relative fp32 on logits, loss and gradient norms.  Logits and
loss meet it directly.  Gradients of this 26-block BN/ReLU network are NOT reproducible to 1e-3
by fp32 arithmetic itself: the reference run in fp32 and the same reference code run in fp64
differ by ~1e-2 in the gradient vector (ReLU decisions of near-zero pre-activations flip under
1e-6 perturbations of the forward pass; measured in tests/golden/make_golden.py and stored in the
fixtures as the *64 entries).  So gradient criteria are stated against the fp64 value of the
reference with the reference's own fp32-vs-fp64 discrepancy as the noise floor:

    err(candidate vs ref64)  <=  RTOL + K * err(ref32 vs ref64)

with K = 3.  The floor stored in a fixture is ONE draw of that rounding noise (the reference's
fp32 run) and the candidate's error is another, independent draw of the same process, so their
ratio scatters (observed 0.5 ... 2.5 across the fixtures and across kernel revisions that only
changed summation order); K = 3 bounds it without hiding a real defect, which shows up as an
error orders of magnitude above the floor (and in the kernel- and block-level tests).

Round 3: ONE draw underestimates the noise of the small-clip fixtures badly.  tests/golden/probe_conditioning.py perturbs
a fixture's clip by 1e-6 (relative, random) and differentiates the fp64 oracle: the GLOBAL gradient norm of
train_M_2x4x32_s1 then moves by 1.3e-2 (median of six draws; 3.2e-2 at most), of train_M_2x4x111_s1 by up to 2.2e-3, of
train_M_2x4x158_s2 -- the fixture that went red on the driver's box in round 2 with 6.4e-3 -- by up to 1.1e-3, single tensors
by 3-26 %: one ReLU unit that switches carries that much weight when the batch statistics of stage 4 span 8-200 voxels.  No
fp32 evaluation order reproduces such a fixture to 1e-3 except by luck (a reordering of two reduction epilogues of the
channelwise kernels moved train_M_2x4x32_s1 from 5.0e-5 to 2.3e-3; so did changing a tile height), and the
reference's own fp32 draw stored in the fixture is one lucky sample.  tests/golden/conditioning.json holds the probe's result
per fixture; a fixture whose MEDIAN probed response (six draws) exceeds 0.3 x RTOL gets `COND_K` x that median added to its
bounds (round 3: one worst-tensor allowance for every tensor -- 12 ... 95 %, too wide; round 4: see check_grads -- the
global response on the whole-vector criteria, the probed response of EACH tensor, capped, on that tensor, none at all when
the bitwise result is pinned by tests/golden/grad_hashes.json, and the two worst fixtures dropped).  The BASELINE
shape 8 x 16 x 224^2 (median 2.0e-4) and train_M_16x2x47_s4 stay below the threshold and keep the plain bounds; the
batch-64 / 128 fixtures are too large to probe on this container's CPU and keep the plain bounds as well.
"""
import numpy as np

RTOL = 1e-3
COND_K = 3.0


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def check_forward(logits, loss, g, rtol=RTOL):
    """logits [B, n_classes] and scalar loss against the fp32 reference."""
    e = rel(logits, g["logits"])
    assert e < rtol, "logits rel err %.3e" % e
    el = abs(float(loss) - float(g["loss"])) / abs(float(g["loss"]))
    assert el < rtol, "loss rel err %.3e" % el
    return e, el


def conditioning(case):
    """The probe's record of a fixture (tests/golden/conditioning.json), or None."""
    import json
    import os
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "conditioning.json")
    if not os.path.exists(p):
        return None
    return json.load(open(p)).get(case)


# fixtures whose gradient-vs-reference criteria are DROPPED unless their bitwise result is pinned (round 4, VERDICT r03 item 4
# iii): the probe's median response is 1.3e-2 (2x4x32: stage 4 sees 8 voxels per channel) / 3.7e-3 with single tensors
# moving by 27-44 % (L at 4x4x96^2, 55 blocks): an allowance that wide asserts nothing.  Logits, loss, BN statistics,
# finiteness of every gradient and the committed hash (tests/test_determinism_gpu.py) are still checked.
GRAD_CRITERIA_DROPPED = ("train_M_2x4x32_s1", "train_L_4x4x96_s1")
COND_CAP = 0.1          # largest per-tensor allowance (10 x ... a tensor the probe moves by more than 3 % is not asserted tighter)


def grad_report(grads, g, sketch_fn, rtol=RTOL):
    """Every gradient error measure of a candidate against a fixture, with the fixture's own noise floors and the ratio of
    each error to its PLAIN bound (rtol + 3 x floor; the global norm: rtol, or rtol + 3 x floor where the reference's own
    fp32 run is further than rtol / 2 from its fp64 value).  Nothing is asserted here."""
    names = list(g["grad_names"])
    assert names == list(grads.keys()), "parameter name/order mismatch"
    n64, n32 = g["grad_norms64"], g["grad_norms"]
    got = np.array([float(np.linalg.norm(np.asarray(grads[k], dtype=np.float64))) for k in names])
    rep = {}
    gg = np.sqrt((got ** 2).sum())
    floor = abs(float(g["grad_global_norm"]) - float(g["grad_global_norm64"])) / float(g["grad_global_norm64"])
    two = "grad_global_norm_draw2" in g.files       # fixture with a second fp32 draw of the reference: floor = the larger
    if two:
        floor = max(floor, abs(float(g["grad_global_norm_draw2"]) - float(g["grad_global_norm64"])) / float(g["grad_global_norm64"]))
    rep["global_norm_err"] = abs(gg - float(g["grad_global_norm64"])) / float(g["grad_global_norm64"])
    rep["global_norm_floor"] = floor
    rep["global_norm_bound"] = rtol if floor < 0.5 * rtol else rtol + 3 * floor
    scale = n64 + 1e-6 * float(g["grad_global_norm64"])
    e_got = np.abs(got - n64) / scale
    e_ref = np.abs(n32 - n64) / scale
    if two:
        e_ref = np.maximum(e_ref, np.abs(g["grad_norms_draw2"] - n64) / scale)
    rep["_names"], rep["_e_got"], rep["_e_ref"] = names, e_got, e_ref
    rep["norm_err_median"], rep["norm_floor_median"] = float(np.median(e_got)), float(np.median(e_ref))
    rep["norm_err_max"], rep["norm_floor_max"] = float(np.nanmax(e_got)), float(e_ref.max())
    nonfinite = [names[i] for i in range(len(names)) if not np.isfinite(got[i])]
    if nonfinite:
        rep["nonfinite"] = nonfinite[:12]
    order = np.argsort(-np.nan_to_num(e_got, nan=np.inf))[:8]
    rep["worst_norms"] = ["%s:%.2e(floor %.2e)" % (names[i], e_got[i], e_ref[i]) for i in order]
    sk = sketch_fn(grads)
    rep["sketch_err"] = rel(sk, g["grad_sketch64"])
    rep["sketch_floor"] = rel(g["grad_sketch"], g["grad_sketch64"])
    if two:
        rep["sketch_floor"] = max(rep["sketch_floor"], rel(g["grad_sketch_draw2"], g["grad_sketch64"]))
    full = {}
    for k in g.files:
        if k.startswith("grad64/"):
            name = k[7:]
            e = rel(grads[name], g[k])
            f = rel(g["grad/" + name], g[k])
            if two:
                f = max(f, rel(g["grad_draw2/" + name], g[k]))
            full[name] = (e, f)
    rep["_full"] = full
    # ratios to the PLAIN bounds (what a pinned result is held to)
    rep["plain_ratio_global"] = rep["global_norm_err"] / rep["global_norm_bound"]
    rep["plain_ratio_norm_max"] = rep["norm_err_max"] / (rtol + 3 * rep["norm_floor_max"])
    rep["plain_ratio_norm_median"] = rep["norm_err_median"] / (rtol + 3 * rep["norm_floor_median"])
    rep["plain_ratio_sketch"] = rep["sketch_err"] / (rtol + 3 * rep["sketch_floor"])
    rep["plain_ratio_full"] = max([e / (rtol + 3 * f) for e, f in full.values()] or [0.0])
    return rep


def check_grads(grads, g, sketch_fn, rtol=RTOL, cond=None, pinned=False, case=None):
    """grads: {name: array-like}.  Returns a dict of measured errors (for reporting).  The WHOLE report -- global norm,
    per-parameter norms with the names of the worst few, sketch, the small gradients shipped in full -- is computed before
    anything is asserted, and every assertion message carries it: a red record must localise the wrong tensor.

    pinned = True: the bitwise result of this case on the current kernel sources is committed (tests/golden/grad_hashes.json)
    and tests/test_determinism_gpu.py has compared it -- the result is the one whose margins were recorded, so the PLAIN
    bounds apply with no conditioning allowance.  pinned = False (kernel sources without a committed hash, the CPU oracle):
    ill-conditioned fixtures (module docstring) get COND_K x the probe's MEDIAN response -- the global response on the
    global-norm, median-norm and sketch criteria (whole-vector measures), the response of THAT tensor, capped at
    COND_CAP, on the per-tensor criteria; tensors the probe did not move keep the plain bound."""
    rep = grad_report(grads, g, sketch_fn, rtol)
    names, e_got, e_ref, full = rep.pop("_names"), rep.pop("_e_got"), rep.pop("_e_ref"), rep.pop("_full")
    fails = []
    if "nonfinite" in rep:
        fails.append("%d parameters with non-finite gradients" % len(rep["nonfinite"]))
    ill = (not pinned) and bool(cond) and float(cond.get("median_global", 0.0)) > 0.3 * rtol
    cg = float(cond["median_global"]) if ill else 0.0
    per = (cond.get("tensor_median") or {}) if ill else {}
    allow = lambda name: min(COND_K * float(per.get(name, 0.0)), COND_CAP)
    rep["cond_global"], rep["pinned"] = cg, bool(pinned)
    rep["cond_tensors_allowed"] = len([n for n in names if allow(n) > 0])
    if (case in GRAD_CRITERIA_DROPPED) and not pinned:
        rep["gradient_criteria"] = "dropped (ill-conditioned fixture, unpinned sources)"
        assert not fails, (fails, rep)
        return rep
    if not rep["global_norm_err"] < max(rep["global_norm_bound"], rtol + COND_K * cg):
        fails.append("global norm: %.3e >= %.3e" % (rep["global_norm_err"], max(rep["global_norm_bound"], rtol + COND_K * cg)))
    if not rep["norm_err_median"] <= rtol + 3 * rep["norm_floor_median"] + COND_K * cg:
        fails.append("median per-parameter norm error")
    bad = [(names[i], e_got[i]) for i in range(len(names))
           if not e_got[i] <= rtol + 3 * rep["norm_floor_max"] + allow(names[i])]
    if bad:
        fails.append("per-parameter norm error: " + ", ".join("%s %.2e" % b for b in bad[:6]))
    if not rep["sketch_err"] <= rtol + 3 * rep["sketch_floor"] + COND_K * cg:
        fails.append("sketch")
    worst, worst_name = 0.0, None
    for name, (e, f) in full.items():
        b = rtol + 3 * f + allow(name)
        if not e / b <= worst:
            worst, worst_name = e / b, name
        if not e <= b:
            fails.append("full gradient %s: %.3e > %.3e" % (name, e, b))
    rep["full_grad_worst_ratio"] = worst
    rep["full_grad_worst"] = worst_name
    assert not fails, (fails, rep)
    return rep


def fmt(rep):
    """One-line rendering of a check_grads report."""
    return " ".join(("%s=%.2e" % (k, v)) if isinstance(v, (float, np.floating)) else ("%s=%s" % (k, v))
                    for k, v in sorted(rep.items()))


def check_bn_stats(new_stats, g, rtol=RTOL):
    for k in g.files:
        if k.startswith("rm/"):
            e = rel(new_stats[k[3:] + ".split_bn.running_mean"], g[k])
            assert e < rtol, (k, e)
        if k.startswith("rv/"):
            e = rel(new_stats[k[3:] + ".split_bn.running_var"], g[k])
            assert e < rtol, (k, e)
