python3 scripts/graph_memset_probe.py > gpurun_out/r03_graph_memset_probe.txt 2> gpurun_out/r03_graph_memset_probe.err; echo probe rc=$?; cat gpurun_out/r03_graph_memset_probe.txt; tail -3 gpurun_out/r03_graph_memset_probe.err
timeout -k 10 600 python -m pytest tests/test_gpu_distributed.py -x -q > gpurun_out/r03_t5.log 2>&1; echo rc=$?; tail -4 gpurun_out/r03_t5.log
python3 -c "import __graft_entry__ as g; g.smoke()"
