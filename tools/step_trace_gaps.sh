cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gap
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-vocoder --no-collate --no-kernel-profile > $GRAFT_REPO_ROOT/gpurun_out/prof_gap.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py /tmp/prof_gap | tee gpurun_out/trace_gaps.json
