#!/bin/bash
# Round-3 GPU pass i: the N > 1 code path of bench.py on one GPU (RCCL init, per-rank all_gather, output gather), smoke(), bench line.
set -u
TAG=${1:-r03i}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
S2M2_BENCH_FORCE_GATHER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_force_gather.json 2> $OUT/bench_force_gather.err; echo "force-gather rc=$?"
tail -3 $OUT/bench_force_gather.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_N1.json 2>/dev/null; echo "bench rc=$?"
python - <<PY
import json
for n in ("bench_force_gather", "bench_N1"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, round(d["value"], 2), "pairs/s", d["per_rank_ms_per_step"], d["roofline"]["variant"], round(d["roofline"]["frac"], 3), d.get("secondary", {}).get("value"))
PY
