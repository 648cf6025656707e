T=tests/test_harness_gpu.py
run() { python -m pytest "$@" -x -q -m gpu > gpurun_out/fl.log 2>&1; echo "rc=$? :: $*"; }
run $T::test_checkpoint_round_trip_on_device
run $T -k "time_script or checkpoint"
run $T -k "experiment or checkpoint"
run $T -k "experiment or time_script"
run $T -s
run $T -v
run $T
