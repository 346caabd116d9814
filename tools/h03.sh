cd /root/repo
for lv in 1000 3000; do
PFQ_BENCH_PARITY_READS=200 timeout -k 10 150 python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --leaves $lv --nbits 11981322 --hashes 17 --k 20 --read-len 100 --threshold 0.3 --reads-per-step 2097152 > gpurun_out/h03_$lv.json 2> gpurun_out/h03_$lv.err; echo "leaves $lv rc=$?"; python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/h03_$lv.json") if l.startswith("{")][-1]); print(d["value"]/1e6, d["kernel_ms_per_step"], d.get("fallback_pairs"))
except Exception as e: print("no line", e)
PY
done
