cd /root/repo
python -m pytest tests -m gpu -x -q > gpurun_out/h03d_pytest.log 2>&1; rc=$?; tail -3 gpurun_out/h03d_pytest.log; [ $rc -eq 0 ] || exit 1
PFQ_BENCH_PARITY_READS=200 timeout -k 10 150 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --leaves 3000 --nbits 11981322 --hashes 17 --k 20 --read-len 100 --threshold 0.3 --reads-per-step 4194304 > gpurun_out/h03d_proxy.json 2> gpurun_out/h03d_proxy.err
PFQ_BENCH_PARITY_READS=200 timeout -k 10 150 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --leaves 1000 --nbits 11981322 --hashes 17 --k 20 --read-len 150 --threshold 0.3 --reads-per-step 4194304 > gpurun_out/h03d_proxy150.json 2> gpurun_out/h03d_proxy150.err
python - <<PY
import json
for n in ("proxy","proxy150"):
    try:
        d=json.loads([l for l in open("gpurun_out/h03d_%s.json"%n) if l.startswith("{")][-1]); print(n, round(d["value"]/1e6,1), {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d.get("candidates_last_step"), d.get("hits_last_step"))
    except Exception as e: print(n, "no line", e)
PY
bash tools/gpu_scen.sh h03d t03e t07e
