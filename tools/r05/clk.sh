cd /root/repo
python tools/debug/quick_bench.py bfloat16 256 768 400 > /tmp/qb.log 2>&1 &
PID=$!
sleep 20
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 1; done
wait $PID
tail -2 /tmp/qb.log
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -v "^=" | head
