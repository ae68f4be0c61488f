for g in 2 3 1; do
HX_EXTRA_FLAGS_HX_SIM="-DHX_PT_GROUP=$g" python -c "from isaac_amd import build; build.build(force=True)" > /dev/null 2>&1
echo "PT_GROUP=$g"; python tools/sim_bench.py 4096 300 trimesh 2>/dev/null | tail -1; python tools/sim_bench.py 4096 300 plane 2>/dev/null | tail -1
done
python -c "from isaac_amd import build; build.build(force=True)" > /dev/null 2>&1
