#!/bin/bash
# quick parity probe, the whole GPU suite, single-stream bench, kernel trace
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/full; mkdir -p $O
timeout -k 10 300 python tests/manual/gpu_quick.py > $O/quick.log 2>&1 || { tail -30 $O/quick.log; exit 1; }
tail -1 $O/quick.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 200 python bench.py --streams 1 --images-per-launch 1 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_s1.json 2> $O/bench_s1.err || { tail -20 $O/bench_s1.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/bench_s1.json")); r=d["roofline"]
print("value %.0f  transform %.2f  entropy %.2f  pack %.2f  total %.2f  %s" % (d["value"], r["kernel_us"], r["entropy_us"], r["pack_us"], r["sum_kernels_us"], d["parity"]))
PY
bash tools/gpu_trace.sh > $O/trace.txt 2>&1; cat $O/trace.txt | grep -v rocclr
