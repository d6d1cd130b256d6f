#!/bin/bash
# Same-box A/B by rocprofv3 kernel trace: the shipped library and build_variants/lib_<name>.so for every name given,
# ROUNDS (default 2) interleaved rounds, at IPLS images per launch (default "8 1").  Prints mean kernel durations.
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r3ab}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 python3 $R/tests/manual/gpu_quick.py > $O/quick.log 2>&1 || { tail -30 $O/quick.log; exit 1; }
for v in "$@"; do
  JPEGAMD_LIB=$R/build_variants/lib_$v.so timeout -k 10 300 python3 $R/tests/manual/gpu_quick.py > $O/quick_$v.log 2>&1 || { echo "variant $v: parity FAILED"; tail -20 $O/quick_$v.log; }
done
for round in $(seq 1 ${ROUNDS:-2}); do
for v in default "$@"; do
  L=$R/jpeg-image-compression_amd/libjpegamd.so; [ $v != default ] && L=$R/build_variants/lib_$v.so
  for ipl in ${IPLS:-8 1}; do
    steps=40; [ $ipl = 1 ] && steps=150
    JPEGAMD_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$v.$ipl.$round -o t --output-format csv -- python3 $R/bench.py --streams 1 --images-per-launch $ipl --steps $steps --warmup 10 --no-cpu-baseline --no-one-image-pass $BENCH_EXTRA > $O/$v.$ipl.$round.json 2> $O/$v.$ipl.$round.err || [ $? -eq 3 ] || { tail -20 $O/$v.$ipl.$round.err; exit 1; }
    python3 - <<PY
import csv
from collections import defaultdict
d=defaultdict(list)
for r in csv.DictReader(open("$O/$v.$ipl.$round/t_kernel_trace.csv")):
    n=r["Kernel_Name"].split("(")[0].replace("jpegamd::","").replace("void ","").split("<")[0]
    d[n].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
m={k:sum(x)/len(x)/1e3 for k,x in d.items() if len(x)>20}
tot=sum(m.values())
print("%-10s ipl $ipl round $round: " % "$v" + "  ".join("%s %.2f" % (k.replace("k_",""), x) for k,x in m.items()) + "  | sum %.2f  per image %.2f" % (tot, tot/$ipl))
PY
  done
done
done
