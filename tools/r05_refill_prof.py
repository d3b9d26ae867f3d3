#!/usr/bin/env python3
"""Phase clocks and lane counts of k_thermal_q (ARTIS_AMD_REFILL=1) and of k_thermal, bench workload, one timestep, from a
-DARTIS_PROFILE build (ARTIS_AMD_SO): python tools/r05_refill_prof.py [packets]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from artis_amd import abi, synth, engine
npk = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
model, cs, ts, aux = synth.build("w7", ncoord=50)
pk = synth.make_packets(model, aux, npk, seed_base=1281360349, kpkt_fraction=0.02, seed=99)
for refill in ("0", "1"):
    os.environ["ARTIS_AMD_REFILL"] = refill
    est = abi.estimators_for(model, "classic")
    eng = engine.Engine(model)
    eng.set_cellstate(cs, ts)
    eng.update_packets(pk.copy(), est)
    s = np.asarray(est.stats).astype(float)
    bd = eng.last_kernel_breakdown()
    eng.close()
    tr = s[abi.STAT_X_MA_JUMPS]
    if refill == "1":
        print(f"k_thermal_q: {bd['thermal_ms']:.1f} ms in {bd['thermal_launches']} launches; transitions {tr:.4g}; wave-rounds {s[46]:.4g}; lanes per round {s[45] / max(s[46], 1):.1f}; "
              f"service passes {s[47]:.4g}, slots per pass {s[44] / max(s[47], 1):.1f}; wave clocks: service {16 * s[42]:.4g} ({s[42] / (s[42] + s[43]):.2f}), walk {16 * s[43]:.4g}; "
              f"clocks per wave-round {16 * s[43] / max(s[46], 1):.0f}, per service pass {16 * s[42] / max(s[47], 1):.0f}")
        names = {59: "pull + hot line + context", 60: "the process that ended the walk", 61: "k-packet step", 62: "prepare + store + context", 63: "classify + append + stacks"}
        print("             service pass by part (clocks per pass): " + ", ".join(f"{n} {16 * s[k] / max(s[47], 1):.0f}" for k, n in names.items()))
    else:
        tot = s[42] + s[43] + s[44] + s[45]
        print(f"k_thermal:   {bd['thermal_ms']:.1f} ms in {bd['thermal_launches']} launches; transitions {tr:.4g}; wave-rounds {s[46]:.4g}; lanes per round {tr / max(s[46], 1):.1f}; "
              f"wave clocks: pull+load {s[42] / tot:.2f}, transition phases {s[43] / tot:.2f}, k-packet phases {s[44] / tot:.2f}, store+append {s[45] / tot:.2f}; "
              f"clocks per wave-round {16 * s[43] / max(s[46], 1):.0f}; k-packet wave-iterations {s[47]:.4g}, clocks each {16 * s[44] / max(s[47], 1):.0f}")
