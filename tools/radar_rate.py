"""use_radar_cov throughput (developer figure): lockstep batches of 256 registrations of 2 000-point scans against a 2 M-point map."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan
ctx = Context(0)
world = synth.make_world(2_000_000, seed=1001)
for method in (1, 2, 3):
    m = IcpMethod(method)
    vm = VoxelHashMap(1.0, 30, ctx); vm.AddPoints(world)
    if method >= 2: vm.CalVoxelCovAll()
    if method == 1: vm.CalPointCovAll(0.4)
    scans, T0s = [], []
    for i in range(256):
        sc, Tt = synth.make_scan(world, 2000, seed=10 + i)
        scans.append(Scan(ctx, sc)); T0s.append(synth.perturb(Tt, seed=100 + i, max_trans=0.15, max_rot_deg=0.5))
    for radar in (0, 1):
        reg = Registration(RegistrationConfig(icp_method=m, use_radar_cov=radar), ctx)
        reg.RunRegisterStream(scans, vm, T0s, slots=256)
        t = time.perf_counter()
        for _ in range(5):
            out = reg.RunRegisterStream(scans, vm, T0s, slots=256)
        dt = (time.perf_counter() - t) / 5
        print(m.name, "radar" if radar else "plain", f"{256 / dt:9.0f} registrations/s  iterations {np.mean([o['iterations'] for o in out]):.2f}  success {np.mean([o['is_success'] for o in out]):.2f}")
