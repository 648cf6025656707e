"""Duration of the single-chain K1 launch as rocprofv3 sees it, split by where the launch sits: inside the real training
step (previous dispatch = the gradient pack / another framework kernel; theta and momentum were last touched one whole
forward/backward ago, so they come from the Infinity Cache / HBM, not L2) vs inside bench.py's `roofline` leg (256
back-to-back K1 launches: operands L2-hot). Explains why rocprofv3's AVERAGE over a bench run sits above the HIP-event
figure of the roofline leg.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/k1w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --ref-style-steps 0 --multi-chain-probe 0
    python3 tools/exp/k1_in_workload.py /tmp/k1w gpurun_out/k1_in_workload.json"""
import csv, glob, json, statistics, sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
is_k1 = lambda n: 'k_sgmcmc_step_ctl<false, false>' in n
groups = {'in_training_step': [], 'back_to_back': []}
for i, (s, e, n) in enumerate(rows):
    if not is_k1(n):
        continue
    prev = rows[i - 1][2] if i else ''
    groups['back_to_back' if is_k1(prev) else 'in_training_step'].append((e - s) / 1e3)
out = {}
for k, v in groups.items():
    if v:
        v.sort()
        out[k] = dict(launches=len(v), median_us=round(statistics.median(v), 3), mean_us=round(sum(v) / len(v), 3),
                      p10_us=round(v[len(v) // 10], 3), p90_us=round(v[9 * len(v) // 10], 3),
                      frac_of_8TBps_at_median=round(20 * 273408 / statistics.median(v) / 1e6 / 8, 4))
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(out))
