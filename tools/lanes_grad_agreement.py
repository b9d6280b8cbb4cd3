"""How closely do first-step gradients agree with and without the side streams (fp32 / bf16)?  Prints the worst
parameters; tests/test_lanes_gpu.py bounds them."""
import copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/speech-separation_amd", ROOT + "/tests"):
    sys.path.insert(0, p)
from test_lanes_gpu import _model, _steps
from helpers import rel_err
for prec in ("fp32", "bf16"):
    base = _model(3, prec)
    other, again = copy.deepcopy(base), copy.deepcopy(base)
    _, g0, _ = _steps(base, False, 1, prec)
    _, g1, _ = _steps(other, True, 1, prec)
    _, g2, _ = _steps(again, False, 1, prec)
    rows = sorted(((rel_err(g1[k], g0[k]), rel_err(g2[k], g0[k]), float(g0[k].abs().max()), k) for k in g0), reverse=True)
    print(prec, "worst lanes-vs-none (rel err, floor, |g|max, name):")
    for r in rows[:6]:
        print("   %.2e %.2e %.2e %s" % r)
    sig = [r for r in rows if r[2] > 1e-6]
    print("   worst among |g|max > 1e-6: %.2e %s" % (sig[0][0], sig[0][3]))
