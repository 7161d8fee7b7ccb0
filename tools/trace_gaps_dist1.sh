#!/bin/bash
# Run ON THE MI355X BOX: kernel timeline of the one-rank COLLECTIVE path (RCCL communicator formed, bench.py as torch.distributed.run
# would start it) -> the launch sequence of an ICP iteration and the gaps between launches.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gaps_dist1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29547 timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o b --output-format csv -- \
  python $R/bench.py --gpus 1 --no-cpu --no-extras --warmup 0 --steps 2 --batch 512 > $OUT/bench.json 2> $OUT/err
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY' > $OUT/gaps.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-40:]
    out.append((name, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
idx = [i for i, o in enumerate(out) if "k_accumulate" in o[0]]
mid = idx[len(idx) // 2]
print("# launches %d .. %d of %d (the middle of the timed region): name, duration, gap to the previous launch's end" % (mid, mid + 16, len(out)))
for n, d, g in out[mid:mid + 16]:
    print("%-42s dur %8.1f us  gap %7.1f us" % (n, d, g))
PY
rm -rf $OUT/trace
cat $OUT/gaps.txt
