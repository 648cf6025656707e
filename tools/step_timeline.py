"""Where one minibatch step's time goes, from a rocprofv3 kernel trace: the dispatches between two consecutive
single-chain K1 launches that sit inside training-step replays are one step; per kernel name: launches per step, busy
microseconds per step, and the idle time between dispatches (end of one to start of the next).
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --steps 2 --warmup 1 \
        --no-cpu-baseline --ref-style-steps 0 --multi-chain-probe 0
    python3 tools/step_timeline.py /tmp/tl gpurun_out/step_timeline.json"""
import csv, glob, json, re, statistics, sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
is_k1 = lambda n: 'k_sgmcmc_step_ctl<false, false>' in n
k1 = [i for i, r in enumerate(rows) if is_k1(r[2])]
steps = []
for a, b in zip(k1, k1[1:]):
    n = b - a
    if 40 <= n <= 400 and not is_k1(rows[a + 1][2]):        # a forward/backward lies between them (round 5: ~146 dispatches, round 6: ~75)
        steps.append((a + 1, b + 1))                          # dispatches after K1 a up to and including K1 b
if not steps:
    sys.exit('no training steps found in the trace')
med_len = statistics.median(b - a for a, b in steps)
steps = [s for s in steps if s[1] - s[0] == med_len]          # identical replays only


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:110]


busy, count = defaultdict(float), defaultdict(int)
wall, idle, gaps = [], [], []
for a, b in steps:
    wall.append((rows[b - 1][1] - rows[a - 1][1]) / 1e3)      # end of previous K1 -> end of this K1
    g = 0.0
    for i in range(a, b):
        s, e, n = rows[i]
        busy[short(n)] += (e - s) / 1e3
        count[short(n)] += 1
        gap = max(0, s - rows[i - 1][1]) / 1e3
        g += gap
        gaps.append(gap)
    idle.append(g)
ns = len(steps)
gaps.sort()
out = dict(steps=ns, dispatches_per_step=int(med_len), wall_us_per_step=round(statistics.median(wall), 1),
           idle_us_per_step=round(statistics.median(idle), 1), gap_us_median=round(gaps[len(gaps) // 2], 2),
           gap_us_p90=round(gaps[9 * len(gaps) // 10], 2),
           kernels=[dict(kernel=k, launches_per_step=round(count[k] / ns, 2), busy_us_per_step=round(v / ns, 1),
                         us_per_launch=round(v / count[k], 2)) for k, v in sorted(busy.items(), key=lambda kv: -kv[1])])
out['busy_us_per_step'] = round(sum(busy.values()) / ns, 1)
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != 'kernels'}))
for k in out['kernels'][:40]:
    print('%6.1f us  x%5.1f  %6.2f us each  %s' % (k['busy_us_per_step'], k['launches_per_step'], k['us_per_launch'], k['kernel']))
