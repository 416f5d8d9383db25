# HBM traffic of the dominant kernel, per launch, as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in
# separate counter-only passes (no trace domains next to --pmc), FETCH_SIZE doubled on gfx950.
# Writes profiles/<round>_hbm_traffic.json (read by bench.py for roofline.traffic) via scripts/summarize_traffic.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-round1}
PAIRS=${PAIRS:-64}
ARGS="bench.py --steps 2 --warmup 1 --no-extras --cpu-pairs 0 --pairs-per-gpu $PAIRS"
rm -rf gpurun_out/traffic_fetch gpurun_out/traffic_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic_fetch -- python3 $ARGS > gpurun_out/traffic_fetch.json 2> gpurun_out/traffic_fetch.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic_write -- python3 $ARGS > gpurun_out/traffic_write.json 2> gpurun_out/traffic_write.err &&
python3 scripts/summarize_traffic.py gpurun_out/traffic_fetch gpurun_out/traffic_write $PAIRS ${CONC:-3} > gpurun_out/${ROUND}_hbm_traffic.json &&
cat gpurun_out/${ROUND}_hbm_traffic.json
