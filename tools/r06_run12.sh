mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06_tests_final.log 2>&1; rc=$?; echo "tests rc $rc" >> gpurun_out/r06_tests_final.log
tail -4 gpurun_out/r06_tests_final.log
if [ $rc -ne 0 ]; then exit 1; fi
bash tools/r06_profiles.sh > gpurun_out/r06_profiles.log 2>&1
tail -12 gpurun_out/r06_profiles.log
ls gpurun_out | grep r06_
