# proxy of the harness geometry at theta 0.3: 3000 leaves (two column groups of 64 row words), 2 M reads of 100 bp
cd /root/repo
run() {
PFQ_BENCH_PARITY_READS=200 timeout -k 10 150 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --leaves 3000 --nbits 11981322 --hashes 17 --k 20 --read-len 100 --threshold 0.3 --reads-per-step 4194304 > gpurun_out/h03b_$1.json 2> gpurun_out/h03b_$1.err; python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/h03b_$1.json") if l.startswith("{")][-1]); print("$1", round(d["value"]/1e6,1), {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d.get("candidates_last_step"))
except Exception as e: print("$1 no line", e)
PY
}
run new
cp build/libpfq_old.so phagefilter_amd/libpfq.so
run old
