run() { # name, env...
name=$1; shift
env PFQ_BENCH_NO_GATE=1 "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 $BARGS > gpurun_out/exp_$name.json 2> gpurun_out/exp_$name.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/exp_$name.json") if l.startswith("{")][0]); print("$name", round(d["value"]/1e6,1), {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d.get("fallback_pairs"), "INVALID" if "INVALID" in d else "")
except Exception as e:
    print("$name", "no line", open("gpurun_out/exp_$name.err").read()[-300:])
PY
}
