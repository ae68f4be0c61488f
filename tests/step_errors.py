"""Shared by the CPU (host build) and GPU replays of the reference-generated env fixtures."""
import numpy as np

# Tolerances of one teacher-forced env step (10 substeps of the fp32 articulated-body algorithm against the float64
# joint-space oracle), per (step, robot) pair: `tight` must hold for at least 99 % of the pairs, `loose` for all.  The pairs
# in between are steps in which a contact point crossed its activation threshold (penetration > 0 and force > 0) one substep
# earlier or later in fp32 than in float64 -- a discontinuity of the contact model, not round-off; their count is printed.
STEP_TOL = dict(obs=(2e-4, 0.1), priv=(1e-3, 0.3), rew=(2e-5, 5e-3), tau=(0.05, 8.0), contact=(1.0, 150.0))


def check_step_errors(label, errs, tol=STEP_TOL, frac=0.99):
    report = {}
    for k, (tight, loose) in tol.items():
        e = np.concatenate([np.asarray(x, np.float64).reshape(-1) for x in errs[k]])
        over = int((e > tight).sum())
        report[k] = "median %.1e worst %.1e, %d of %d pairs over %.0e" % (np.median(e), e.max(), over, e.size, tight)
        assert np.quantile(e, frac) <= tight, f"{label}: {k}: {report[k]}"
        assert e.max() <= loose, f"{label}: {k}: {report[k]}"
    print(label, "teacher-forced step errors:", report)


