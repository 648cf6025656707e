#!/bin/bash
# Round-2 evidence for profiles/: rocprofv3 kernel stats of the default bench command, the kernel micro-benchmark,
# and PMC passes (separate runs, no trace domains) for K1 / K2 / K3 / K5.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r02_profiles.sh'
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r02_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 3 --warmup 1 > $out/bench_line_under_rocprof.json 2> $out/bench_under_rocprof.err; echo "bench under rocprof rc=$?"
python3 $R/tools/prof_summary.py /tmp/prof_bench $out/r02_bench_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kb -- python3 $R/tools/kbench.py > $out/kbench.log 2>&1; echo "kbench rc=$?"
python3 $R/tools/prof_summary.py /tmp/prof_kb $out/r02_kbench_kernel_stats.csv > /dev/null
cp $R/gpurun_out/kbench.json $out/r02_kbench.json 2>/dev/null
cd $R
for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=/tmp/pmc_$(echo $grp | tr ' ' '_')
  rm -rf $d
  (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $R/tools/pmc_only.py $out/pmc_manifest.json > /dev/null 2> $out/pmc_last.err); echo "pmc [$grp] rc=$?"
done
python3 tools/pmc_summary2.py $out/pmc_manifest.json $out/r02_pmc.json /tmp/pmc_* > /dev/null; echo "pmc summary rc=$?"
python3 bench.py > $out/r02_bench_line.json 2> $out/bench.err; echo "plain bench rc=$?"
ls -la $out
