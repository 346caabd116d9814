# timing experiments: PFQ_BIN_DEBUG bits (1 no bucket stores, 2 no LDS binning, 4 no record loads); results are wrong on purpose
for dbg in ${@:-0 1 2 4 6 7}; do
PFQ_BENCH_NO_GATE=1 PFQ_BIN_DEBUG=$dbg timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 > gpurun_out/exp_$dbg.json 2> gpurun_out/exp_$dbg.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/exp_$dbg.json") if l.startswith("{")][0]); print($dbg, {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()})
except Exception as e:
    print($dbg, "no line", open("gpurun_out/exp_$dbg.err").read()[-300:])
PY
done
