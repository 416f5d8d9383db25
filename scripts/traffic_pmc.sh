# HBM traffic of one kernel, per launch, as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in
# separate counter-only passes (no trace domains next to --pmc), FETCH_SIZE doubled on gfx950.
#   bash scripts/traffic_pmc.sh NAME KERNEL_SUBSTRING "key=value ..." python3 <program> [args]
# Writes gpurun_out/<ROUND>_<NAME>_traffic.json (copy into profiles/: bench.py reads it for roofline.traffic).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-round6}
NAME=$1; KERNEL=$2; KEYS=$3; shift 3
rm -rf gpurun_out/traffic_fetch_$NAME gpurun_out/traffic_write_$NAME
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/traffic_fetch_$NAME -- "$@" > gpurun_out/traffic_fetch_$NAME.out 2> gpurun_out/traffic_fetch_$NAME.err &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/traffic_write_$NAME -- "$@" > gpurun_out/traffic_write_$NAME.out 2> gpurun_out/traffic_write_$NAME.err &&
python3 scripts/summarize_traffic.py gpurun_out/traffic_fetch_$NAME gpurun_out/traffic_write_$NAME "$KERNEL" $KEYS > gpurun_out/${ROUND}_${NAME}_traffic.json &&
cat gpurun_out/${ROUND}_${NAME}_traffic.json
