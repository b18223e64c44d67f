#!/bin/bash
# Everything the round's DESIGN.md / bench roofline quote, in one GPU call (about 12 minutes):
#   bash scripts/collect_round.sh r03
set -e
TAG=${1:-r03}
python3 scripts/hbm_microbench.py > gpurun_out/${TAG}_hbm_microbench.txt 2>&1
cp gpurun_out/${TAG}_hbm_microbench.txt profiles/
bash scripts/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
bash scripts/collect_k1_bytes.sh $TAG > gpurun_out/${TAG}_k1bytes.log 2>&1
cp gpurun_out/${TAG}_k1_bytes.json profiles/
python3 bench.py > gpurun_out/${TAG}_default_bench.json 2> gpurun_out/${TAG}_default_bench.err
python3 bench.py --force-dist --no-cpu-baseline > gpurun_out/${TAG}_forcedist_rccl_world1_bench.json 2> gpurun_out/${TAG}_forcedist.err
python3 scripts/graph_memset_probe.py > gpurun_out/${TAG}_graph_memset_probe.txt 2>&1 || true
for f in cfg2 cfg2_maxlag512 cfg3 cfg4 cfg5; do cp gpurun_out/${TAG}_${f}_bench.json gpurun_out/${TAG}_${f}_kernel_stats.csv profiles/; done
cp gpurun_out/${TAG}_default_bench.json gpurun_out/${TAG}_forcedist_rccl_world1_bench.json gpurun_out/${TAG}_graph_memset_probe.txt profiles/
tail -1 gpurun_out/${TAG}_default_bench.json
