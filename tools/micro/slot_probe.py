import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from groove_amd import entities as E, projects as PJ, patches as P, abi_types as T
mode = sys.argv[1]
ctx = E.Context(0)
shift = int(mode[1:]) if mode[0] == "s" else 0
dummies = []
for i in range(shift):
    d = E.FmSynth(ctx, (T.FmParams * 8)(*[P.fm_patch(j) for j in range(8)]))
    if mode[0] == "r":
        bus = ctx.bus(256); d.render_mix(bus, 256); ctx.synchronize()
    dummies.append(d)
if mode[0] == "d":   # destroy the dummies before the project
    pass
for d in dummies: d.destroy()
w = "mixed-131072"
V = PJ.WORKLOADS[w]["voices"]
proj = PJ.Project(ctx, w, np.arange(V))
K = 64
bus = ctx.bus(K * PJ.FRAMES)
for rep in range(3):
    proj.reset()
    for k in range(8): proj.step(bus, k * PJ.FRAMES)
    ctx.synchronize()
    t0 = time.perf_counter()
    for k in range(K): proj.step(bus, k * PJ.FRAMES)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
print(f"{mode}: submit {1e3 * (t1 - t0) / K:.4f}  total {1e3 * (t2 - t0) / K:.4f} ms/step", flush=True)
