"""Regenerate tests/golden/macroatom_reference.json from the REFERENCE's own macroatom.h.

Needs /root/reference (this container only). `make -C oracle ref` compiles oracle/ref_harness/ref_macroatom_main.cc
against /root/reference/macroatom.h (included where it lies, never copied) into oracle/_ref/ref_macroatom; this script
evaluates rad_deexcitation_ratecoeff() (macroatom.h:61) over optically thin, thick, inverted and degenerate cases and
stores the results as hex floats.
"""
import itertools
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EV = 1.6021772e-12
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
cases = []
for eps_ev, A, (gu, gl), (nnu, nnl), t in itertools.product(
        (0.05, 1.9, 7.3, 24.6), (1.0e-2, 3.3e5, 6.3e8), ((2.0, 4.0), (9.0, 1.0)),
        ((0.0, 0.0), (1e-3, 5e2), (4.0e4, 1.0e4), (1e2, 3e9)), (1.728e5, 1.728e6, 8.64e6)):
    args = [repr(eps_ev * EV), repr(A), repr(gu), repr(gl), repr(nnu), repr(nnl), repr(t)]
    res = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_macroatom")] + args, text=True).strip()
    cases.append({"epsilon_trans": eps_ev * EV, "A_ul": A, "g_upper": gu, "g_lower": gl, "nn_upper": nnu, "nn_lower": nnl,
                  "t_current": t, "result": res})
with open(os.path.join(HERE, "macroatom_reference.json"), "w") as f:
    json.dump({"function": "rad_deexcitation_ratecoeff (reference macroatom.h:61)", "cases": cases}, f, indent=0)
print("wrote macroatom_reference.json with", len(cases), "cases")
