"""Regenerate tests/golden/gk31_reference.json from the REFERENCE's own gausskronrod.h.

Needs /root/reference (this container only). `make -C oracle ref` compiles oracle/ref_harness/ref_gk31_main.cc against
/root/reference/gausskronrod.h (the reference source is included where it lies, never copied) into oracle/_ref/ref_gk31;
this script runs gauss_kronrod_integrate<31>(f, a, b, 15, tol, &error) -- the call of integrator<31>() behind
select_continuum_nu() (ratecoeff.cc:563) -- for analytic integrands that exercise the adaptive bisection (steps, a
smooth peak, a kink), the reversed and the empty interval, and stores results and error estimates as hex floats.
"""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = []
for tol in (1e-3, 1e-8):
    CASES += [
        (1, 0.5, 3.0, 0.0, 10.0, tol), (1, 0.05, 0.7, 0.0, 37.5, tol), (1, 2.0, 11.0, 1.0, 4.0, tol),
        (2, 1.0, 0.0, 0.0, 20.0, tol), (2, 0.3, 2.5, 0.0, 55.0, tol), (2, 4.0, 0.1, 0.5, 0.50001, tol),
        (3, 0.3, 1.0, 0.0, 1.0, tol), (3, 2.0, 0.0, -1.0, 5.0, tol), (3, 0.5, 0.25, 0.5, 0.5, tol),
        (1, 0.5, 3.0, 10.0, 0.0, tol), (2, 1e-3, 1e-9, 0.0, 1e4, tol), (3, 1e3, 2.0, 0.0, 2e3, tol),
    ]

subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
out = {"call": "gauss_kronrod_integrate<31>(f, a, b, 15, tol, &error) of the reference's gausskronrod.h",
       "integrands": {"1": "exp(-p0*x)*(1+floor(x*p1))", "2": "x*x*exp(-p0*x)/(1+p1*x*x*x)", "3": "sqrt(fabs(x-p0))+p1"},
       "cases": []}
for mode, p0, p1, a, b, tol in CASES:
    args = [os.path.join(ROOT, "oracle", "_ref", "ref_gk31"), str(mode), repr(p0), repr(p1), repr(a), repr(b), repr(tol)]
    res, err = subprocess.check_output(args, text=True).split()
    out["cases"].append({"mode": mode, "p0": p0, "p1": p1, "a": a, "b": b, "tol": tol, "result": res, "error": err})
with open(os.path.join(HERE, "gk31_reference.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote gk31_reference.json with", len(out["cases"]), "cases")
