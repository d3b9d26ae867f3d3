#!/usr/bin/env python3
"""Wave clocks of the stages of one macro-atom transition in k_thermal (bench workload, one timestep).

Build the instrumented library first (it is not one of the shipped builds):
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -ldl \
        -DARTIS_PROFILE -DARTIS_PROFILE_MA -Iinclude -o scratch/libprof.so artis_amd/csrc/artis_engine.hip
then on the GPU box: python tools/stage_clocks.py        (result of round 2: profiles/r02/k_thermal_stage_clocks.txt)
"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ARTIS_AMD_SO"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scratch", "libprof.so")
from artis_amd import abi, synth, engine
model, cs, ts, aux = synth.build("w7", ncoord=int(sys.argv[1]) if len(sys.argv) > 1 else 50)
pk = synth.make_packets(model, aux, 10000000, kpkt_fraction=0.02)
est = abi.estimators_for(model, "classic")
eng = engine.Engine(model)
eng.set_cellstate(cs, ts)
eng.update_packets(pk, est)
s = np.asarray(est.stats).astype(float)
print("transitions", s[abi.STAT_X_MA_JUMPS], "wave-rounds (slot 46)", s[46], "lanes/round", s[abi.STAT_X_MA_JUMPS] / s[46])
print("k_thermal wave clocks /16: pull+load", s[42], "MA phase", s[43], "kpkt phase", s[44], "store+append", s[45], "kpkt wave-iterations", s[47])
wr = s[46]
for name, slot in (("mark cost", 63), ("rates read (stage 1 wait)", 59), ("process drawn", 60), ("direction searched (stage 2)", 61), ("target read (stage 3)", 62)):
    print(f"{name:32s} {s[slot]:.4g} clocks  = {s[slot] / wr:8.1f} per wave-round")
print("MA phase per wave-round", 16 * s[43] / wr)
print(eng.last_kernel_breakdown() if hasattr(eng, "last_kernel_breakdown") else "")
print("do_kpkt wave clocks /16 (slots 56, 57, 41): ion drawn", s[56], "term drawn", s[57], "process", s[41], " per kpkt wave-iteration:",
      [round(16 * s[k] / max(s[47], 1)) for k in (56, 57, 41)])
