# developer ablation: time the cell kernel with parts compiled out (results are wrong, timings only)
for f in "" "-DELM_SKIP_HARD" "-DELM_SKIP_HARD -DELM_SKIP_CANDS" "-DELM_SKIP_HARD -DELM_SKIP_PAIR -DELM_SKIP_REDUCE -DELM_SKIP_STAGE1"; do
  make -C elimaloc_amd/csrc EXTRA="$f" -B 2>&1 | grep -E " error"
  echo "FLAGS: $f"; python tools/kbench.py --batch 256 --slots 128 --steps 5 --iters 3 --term 0 2>&1 | tail -1 | cut -c1-120
done
