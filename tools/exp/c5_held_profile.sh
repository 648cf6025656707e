R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r05k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
URSA_BN_HELD=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5h -- python3 $R/bench.py --config c5 --c5-batch 1024 --steps 3 --warmup 0 --detail-out $out/c5h_detail.json > $out/c5h_line.json 2> $out/c5h.err; echo "c5 held under rocprof rc=$?"
python3 $R/tools/prof_summary.py /tmp/prof_c5h $out/r05_c5_kernel_stats_held_opt_in.csv > /dev/null
grep "k_bn" $out/r05_c5_kernel_stats_held_opt_in.csv | cut -c30-75,150-330 | head -14
