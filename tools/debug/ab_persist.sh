for i in 1 2 3; do
  for v in P N; do
    if [ $v = N ]; then export EEGLDM_GEMM_BIG_NO_PERSIST=1; else unset EEGLDM_GEMM_BIG_NO_PERSIST; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-parts 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
  done
done
