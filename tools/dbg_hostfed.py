#!/usr/bin/env python3
"""developer check: host-fed stream vs resident stream on ragged small scans; prints the registrations that differ"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, Registration, RegistrationConfig, IcpMethod, Scan, VoxelHashMap, PinnedBuffer
ctx = Context(0)
world = synth.make_world(100000, seed=1001)
vm = VoxelHashMap(1.0, 30, ctx); vm.AddPoints(world)
reg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), ctx)
sizes = [6000, 1000, 0, 257, 5000, 3000, 256, 4097, 1, 2500, 7000] * 13
hosts, T0s = [], []
for i, n in enumerate(sizes):
    sc, Tt = synth.make_scan(world, max(n, 1), seed=500 + i)
    hosts.append(sc[:n]); T0s.append(synth.perturb(Tt, seed=600 + i, max_trans=0.05 + 0.004 * i, max_rot_deg=0.02 * (i + 1)))
scans = [Scan(ctx, h) for h in hosts]
want = reg.RunRegisterStream(scans, vm, T0s, slots=9)
pin = PinnedBuffer(sum(h.size for h in hosts))
for name, packed, slots in (("pinned/9", reg.pack_host_inputs(hosts, T0s, pinned=pin), 9), ("pageable/5", reg.pack_host_inputs(hosts, T0s), 5),
                            ("pinned/200", reg.pack_host_inputs(hosts, T0s, pinned=pin), 200)):
    for rep in range(3):
        got = reg.RunRegisterStreamHost(packed, vm, slots=slots)
        bad = [k for k, (a, b) in enumerate(zip(got, want)) if not (np.array_equal(a["T"], b["T"]) and a["iterations"] == b["iterations"])]
        print(name, "rep", rep, "serial" if os.environ.get("ELM_HOSTFED_SERIAL") else "streams", "mismatches:", bad[:40], len(bad), flush=True)
