set -e
cd /root/repo
E=gpurun_out/r05_evidence
part=$1
case $part in
a) for f in r05_conv_wgrad_probe.json r05_conv_wgrad_probe_kernels.txt r05_conv_fwd_probe.json r05_conv_fwd_probe_kernels.txt r05_bn_fused_bench.json r05_bn_held_ab.json r05_bn_held_timeline.txt r05_g16_gate_parity.json r05_grouped_conv_ab.json r05_k1_ctl_bench.json r05_k3_spread.json r05_k5_prefetch_ab.txt r05_kbench.json r05_kbench_kernel_stats.csv r05_ticket_probe.txt; do cp $E/$f profiles/; done
   grep -v amdgpu.ids $E/r05_gate_probe_bisect.log | grep -v "^$" | cut -c1-600 > profiles/r05_gate_probe_bisect.log ;;
b) cp $E/r05_bench_kernel_stats.csv $E/r05_c4_kernel_stats.csv $E/r05_c5_kernel_stats.csv $E/r05_k1_in_workload.json $E/r05_pmc.json profiles/ ;;
c) cp $E/r05_bench_line_stock_conv.json $E/r05_bench_detail_stock_conv.json profiles/; cp $E/r05_bench_line.json $E/r05_bench_line_driver_cmd.json $E/r05_bench_line_stock_bn.json $E/r05_bench_detail.json $E/r05_bench_detail_driver_cmd.json $E/r05_bench_detail_stock_bn.json $E/r05_bench_driver_cmd_wall_time.txt $E/r05_time_script_preresnet20.json $E/r05_experiment_results.csv profiles/ ;;
d) cp $E/r05_c4_bench_line.json $E/r05_c4_bench_detail.json $E/r05_c5_bench_line.json $E/r05_c5_bench_detail.json $E/r05_c5_bench_line_held_opt_in.json profiles/ ;;
e) cp $E/r05_c4_bench_line_held_opt_in.json profiles/ ;;
f) cp $E/r05_step_timeline.json profiles/ ;;
esac
cp $E/r05_sha256_part_$part.txt profiles/r05/
python3 - <<'PY'
import json,glob
for p in glob.glob('/root/repo/profiles/r05_*bench_line*.json'):
    l=open(p).read().strip().splitlines()[-1]; json.loads(l); open(p,'w').write(l+'\n')
PY
echo copied $part
