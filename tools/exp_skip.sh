for f in "" "-DELM_SKIP_HARD" "-DELM_SKIP_HARD -DELM_SKIP_PAIR" "-DELM_SKIP_HARD -DELM_SKIP_PAIR -DELM_SKIP_REDUCE" "-DELM_SKIP_HARD -DELM_SKIP_PAIR -DELM_SKIP_REDUCE -DELM_SKIP_STAGE1"; do
  make -C elimaloc_amd/csrc EXTRA="$f" -B 2>&1 | grep -E " error" 
  echo "FLAGS: $f"; ELM_KERNEL=cell python tools/kbench.py --batch 16 --steps 10 --iters 3 --term 0 2>&1 | tail -1
done
