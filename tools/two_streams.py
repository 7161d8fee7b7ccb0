#!/usr/bin/env python3
"""developer experiment: do two concurrent half-streams (two contexts, two HIP streams, 64 slots each) beat one stream of 128 slots?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap
B = int(os.environ.get("TS_BATCH", "2048"))
world = synth.make_world(10_000_000, seed=1001)
ctxs = [Context(0), Context(0)]
vms = []
for c in ctxs:
    vm = VoxelHashMap(1.0, 30, c); vm.AddPoints(world); vms.append(vm)
regs = [Registration(RegistrationConfig(icp_method=IcpMethod.P2P), c) for c in ctxs]
from concurrent.futures import ThreadPoolExecutor
def gen(i):
    sc, Tt = synth.make_scan(world, 131072, seed=2002 + i)
    return sc, synth.perturb(Tt, seed=3003 + i)
synth.make_scan(world, 16, seed=1)
with ThreadPoolExecutor(16) as pool:
    data = list(pool.map(gen, range(B)))
# one stream, 128 slots, all B
scans0 = [Scan(ctxs[0], d[0]) for d in data]
p0 = regs[0].pack_inputs(scans0, [d[1] for d in data])
for _ in range(2): regs[0].RunRegisterStream(p0[0], vms[0], p0[1], slots=128, raw=True)
t = time.perf_counter()
for _ in range(5): regs[0].RunRegisterStream(p0[0], vms[0], p0[1], slots=128, raw=True)
t1 = time.perf_counter() - t
print("one stream x128 slots: %.0f reg/s" % (5 * B / t1), flush=True)
# two contexts, each half of the batch, 64 slots each, concurrently
halves = []
for h in range(2):
    sc = [Scan(ctxs[h], d[0]) for d in data[h::2]] if h == 1 else scans0[0::2]
    halves.append(regs[h].pack_inputs(sc, [d[1] for d in data[h::2]]))
def run(h, slots, reps):
    for _ in range(reps): regs[h].RunRegisterStream(halves[h][0], vms[h], halves[h][1], slots=slots, raw=True)
for slots in (64, 128):
    for h in range(2): run(h, slots, 1)
    th = [threading.Thread(target=run, args=(h, slots, 5)) for h in range(2)]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    t2 = time.perf_counter() - t
    print("two streams x%d slots: %.0f reg/s" % (slots, 5 * B / t2), flush=True)
