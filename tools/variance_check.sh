for cfg in "50 10" "500 100" "50 10" "500 100" "50 10" "500 100"; do set -- $cfg
timeout -k 10 300 python bench.py --legs none --steps $1 --warmup $2 --no-cpu-baseline --check-rows 0 > gpurun_out/r3M.json 2> gpurun_out/r3M.err || exit 1
python - $1 $2 <<'PY'
import json,sys
d=json.load(open('gpurun_out/r3M.json'))
print('steps',sys.argv[1],'warmup',sys.argv[2],'value %.0f ms_per_step %.4f kernel_ms %.4f p10 %.4f p90 %.4f' % (d['value'], d['ms_per_step'],d['roofline']['kernel_ms'], d['step_ms_device']['p10'], d['step_ms_device']['p90']))
PY
done
