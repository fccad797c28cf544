python -m pytest tests/test_gpu_gn_pipe.py -q 2>&1 | tail -3
python tools/debug/gn_pipe_bench.py 2>&1 | tail -7
for i in 1 2; do
 echo "--- default (pipe for no-addend)"; python bench.py --no-parts --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
 echo "--- no pipe"; EEGLDM_GN_NO_PIPE=1 python bench.py --no-parts --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
 echo "--- pipe incl addend"; EEGLDM_GN_PIPE_ADDEND=1 python bench.py --no-parts --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['final_loss'])"
done
