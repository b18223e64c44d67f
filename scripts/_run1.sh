bash scripts/collect_profiles.sh r03 > gpurun_out/r03_collect.log 2>&1; echo collect rc=$?
bash scripts/collect_k1_bytes.sh r03 > gpurun_out/r03_k1bytes.log 2>&1; echo k1bytes rc=$?
python3 bench.py > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err; echo bench rc=$?
python3 bench.py --force-dist --no-cpu-baseline > gpurun_out/r03_forcedist_bench.json 2> gpurun_out/r03_forcedist.err; echo forcedist rc=$?
python3 scripts/time_pcie.py > gpurun_out/r03_time_pcie.txt 2>&1; echo pcie rc=$?; tail -5 gpurun_out/r03_time_pcie.txt
