R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r05j
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for variant in "parity:--steps 1 --warmup 0 --no-cpu-baseline --ref-style-steps 0 --multi-chain-sweep= --bma-members 0" "noparity:--steps 1 --warmup 0 --no-parity --no-cpu-baseline --ref-style-steps 0 --multi-chain-sweep= --bma-members 30"; do
  tag=${variant%%:*}; args=${variant#*:}
  rm -rf /tmp/pn_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn_$tag -- python3 $R/bench.py $args --detail-out $out/d_$tag.json > $out/line_$tag.json 2> $out/err_$tag.txt; echo "$tag rc=$?"
  python3 $R/tools/prof_summary.py /tmp/pn_$tag $out/stats_$tag.csv > /dev/null
  echo "== $tag: naive / fallback rows"; grep -i "naive\|batched_gemm_xdl" $out/stats_$tag.csv | cut -c1-120
done
