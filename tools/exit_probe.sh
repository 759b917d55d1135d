cd $GRAFT_REPO_ROOT
A="--steps 2 --warmup 1 --presc-steps 16 --cpu-seconds 0 --no-verify --no-boundary --no-overlap-leg --min-seconds 0.01 --scale 20"
echo "== plain"; python bench.py $A > /dev/null 2> gpurun_out/ep1.err; echo rc=$?; tail -2 gpurun_out/ep1.err
echo "== markers off"; LEGION_MARKERS=0 python bench.py $A > /dev/null 2> gpurun_out/ep2.err; echo rc=$?; tail -2 gpurun_out/ep2.err
echo "== sysfs only"; LEGION_LINK_SOURCE=2 python bench.py $A > /dev/null 2> gpurun_out/ep3.err; echo rc=$?; tail -2 gpurun_out/ep3.err
echo "== both off"; LEGION_MARKERS=0 LEGION_LINK_SOURCE=2 python bench.py $A > /dev/null 2> gpurun_out/ep4.err; echo rc=$?; tail -2 gpurun_out/ep4.err
cd /tmp; export TMPDIR=/tmp
echo "== rocprof plain"; timeout -k 5 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/ep5 -- python3 $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/ep5.err; echo rc=$?; grep -i "corrupt\|signal" $GRAFT_REPO_ROOT/gpurun_out/ep5.err | head -3
echo "== rocprof markers off"; LEGION_MARKERS=0 timeout -k 5 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/ep6 -- python3 $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/ep6.err; echo rc=$?; grep -i "corrupt\|signal" $GRAFT_REPO_ROOT/gpurun_out/ep6.err | head -3
echo "== rocprof sysfs"; LEGION_LINK_SOURCE=2 timeout -k 5 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/ep7 -- python3 $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/ep7.err; echo rc=$?; grep -i "corrupt\|signal" $GRAFT_REPO_ROOT/gpurun_out/ep7.err | head -3
