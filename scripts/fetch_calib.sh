# FETCH_SIZE calibration on the access shapes of the ICP kernel (scripts/fetch_calib.hip): one counter-only pass for
# FETCH_SIZE, one for the raw fabric read-request counters.  Writes gpurun_out/<ROUND>_fetch_calibration.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROUND=${ROUND:-round3}
rm -rf gpurun_out/calib_fetch gpurun_out/calib_rdreq
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_fetch -- scripts/fetch_calib > gpurun_out/calib_fetch.out 2> gpurun_out/calib_fetch.err &&
{ rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/calib_rdreq -- scripts/fetch_calib > gpurun_out/calib_rdreq.out 2> gpurun_out/calib_rdreq.err || echo "RDREQ pass failed (counter names?)"; } &&
python3 scripts/fetch_calib_summary.py gpurun_out/calib_fetch gpurun_out/calib_rdreq > gpurun_out/${ROUND}_fetch_calibration.json && cat gpurun_out/${ROUND}_fetch_calibration.json
