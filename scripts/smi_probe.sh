mkdir -p gpurun_out/r2
(for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo; sleep 0.5; done) > gpurun_out/r2/smi.txt &
SMI=$!
python bench.py --steps 40 --warmup 3 --no-cpu-baseline | cut -c1-160
kill $SMI 2>/dev/null
wait $SMI 2>/dev/null
true
