cd /root/repo
run() {
PFQ_BENCH_PARITY_READS=200 timeout -k 10 150 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --leaves 3000 --nbits 11981322 --hashes 17 --k 20 --read-len 100 --threshold 0.3 --reads-per-step 4194304 > gpurun_out/h03c_$1.json 2> gpurun_out/h03c_$1.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/h03c_$1.json") if l.startswith("{")][-1]); print("$1 proxy", round(d["value"]/1e6,1), {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d.get("candidates_last_step"))
except Exception as e: print("$1 no line", e)
PY
PFQ_BENCH_READ_ERRORS=0.01 timeout -k 10 150 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --threshold 0.3 > gpurun_out/h03c_t03e_$1.json 2> gpurun_out/h03c_t03e_$1.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/h03c_t03e_$1.json") if l.startswith("{")][-1]); print("$1 t03e ", round(d["value"]/1e6,1), {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d.get("candidates_last_step"))
except Exception as e: print("$1 no line", e)
PY
}
run cur
for v in v1 v3 v4; do cp build/libpfq_$v.so phagefilter_amd/libpfq.so; run $v; done
