#!/usr/bin/env python3
"""Per (sweep, tile) visit of a tiled run: launches, summed kernel ms, packets of the first launches.
   ARTIS_AMD_CACHE_BUDGET_MB=25000 ARTIS_AMD_TRACE=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 >/dev/null | python tools/tiling_trace.py"""
import re, sys
visits = []
for line in sys.stdin:
    m = re.search(r"sweep (\d+) tile (\d+) of (\d+)", line)
    if m:
        visits.append({"sweep": int(m.group(1)), "tile": int(m.group(2)), "launches": 0, "ms": 0.0, "first": None, "kinds": {}})
        continue
    m = re.search(r"launch \d+ (kind (\d+)|tail) n=(\d+)(?:\+\d+)? ([\d.]+) ms", line)
    if m and visits:
        v = visits[-1]
        v["launches"] += 1
        v["ms"] += float(m.group(4))
        k = m.group(2) or "tail"
        v["kinds"][k] = v["kinds"].get(k, 0.0) + float(m.group(4))
        if v["first"] is None:
            v["first"] = int(m.group(3))
tot = sum(v["ms"] for v in visits)
print(f"{len(visits)} tile visits, {sum(v['launches'] for v in visits)} launches, {tot:.0f} ms in kernels")
for v in visits:
    print(f"sweep {v['sweep']:2d} tile {v['tile']:2d}: {v['launches']:4d} launches {v['ms']:8.1f} ms  first list {v['first']}  by kind {{" +
          ", ".join(f"{k}: {t:.0f}" for k, t in sorted(v['kinds'].items())) + "}")
