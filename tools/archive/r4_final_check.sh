python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
bash tools/box_probe.sh gpurun_out/r4o/box.json > /dev/null 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4o/bench_cfg2_driver_form.json 2> gpurun_out/r4o/bench.err
python tools/fingerprint_table.py gpurun_out/r4o | tail -1 | cut -c1-260
