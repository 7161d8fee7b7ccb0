import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from elimaloc_amd import synth
from elimaloc_amd.registration import Context
from elimaloc_amd.deskew import PcmDeskew
from oracle import oracle
ctx = Context(0)
tot = bad = 0
for seed in range(41, 61):
    st = synth.make_deskew_stream(131072, seed=seed, yaw_rate=0.3 + 0.2 * (seed % 7))
    dk = PcmDeskew(ctx)
    imu = np.concatenate([st["imu_t"][:, None], st["imu_w"]], axis=1)
    ok, out = dk.DeskewPointCloud(st["xyz"], st["time"], st["stamp"], imu, st["odom"])
    front = float(st["time"][0]); scan_end = st["stamp"]; scan_cur = scan_end + front
    iok, itime, irot = oracle.imu_deskew_info(st["imu_t"], st["imu_w"], scan_cur, scan_end)
    ook, inc = oracle.odom_deskew_info(st["odom"], scan_cur, scan_end)
    ref = oracle.deskew_points(st["xyz"], st["time"] - np.float32(front), itime, irot, scan_cur, scan_end, inc)
    tot += out.size; bad += int((out != ref).sum())
print("values", tot, "differing", bad, "max abs diff", float(np.abs(out - ref).max()))
