#!/bin/bash
# BASELINE configs[3] and [4] at full size through bench.py, then short profiled runs for the per-kernel stats.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/run_c4_c5.sh'
out=gpurun_out/r02_c4c5
mkdir -p $out
python bench.py --config c4 > $out/c4.json 2> $out/c4.err; echo "c4 rc=$?"; tail -c 600 $out/c4.json; echo
python bench.py --config c5 --c5-batch 1024 > $out/c5.json 2> $out/c5.err; echo "c5 rc=$?"; tail -c 600 $out/c5.json; echo
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$out/prof_c4 -o c4 --output-format csv -- python3 $R/bench.py --config c4 --steps 2 --warmup 1 --c4-epochs 2 --c4-train 5120 > $R/$out/c4_prof.json 2> $R/$out/c4_prof.err; echo "c4 prof rc=$?"
rocprofv3 --kernel-trace --stats -d $R/$out/prof_c5 -o c5 --output-format csv -- python3 $R/bench.py --config c5 --c5-batch 1024 --steps 3 --warmup 0 > $R/$out/c5_prof.json 2> $R/$out/c5_prof.err; echo "c5 prof rc=$?"
cd $R
find $out -name "*kernel_stats.csv" | head; find $out -name "*_kernel_trace.csv" -size +20M -delete
