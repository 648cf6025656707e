"""Launch each hot kernel a few times in isolation, for rocprofv3 --pmc passes (tools/r02_profiles.sh), and
write the launch manifest (kernel-name pattern, algorithmic bytes per launch) next to the counters.
    python3 tools/pmc_only.py <manifest.json>"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursabench_amd import _native
K = _native.default_kernels()
manifest = []


def note(pattern, label, alg_bytes, **kw):
    manifest.append(dict(pattern=pattern, label=label, algorithmic_bytes_per_launch=alg_bytes, **kw))


# `python3 tools/pmc_only.py <manifest.json> k6_workload`: only K6 at the workload's own first / last stage (their launch
# grids coincide with the roofline-sized shapes' below, and the summary keys counters by kernel + grid: a pass of their own)
K6_ONLY = len(sys.argv) > 2 and sys.argv[2] == 'k6_workload'
K6_SHAPES = ((((128, 16, 32, 32), False), ((128, 64, 8, 8), True)) if K6_ONLY
             else (((1024, 64, 32, 32), False), ((512, 640, 8, 8), True)))
if not K6_ONLY:
    # K1 at the PreResNet-20 arena size through the control-block entry point (the workload's launch)
    n = 273408
    th, g, m = (torch.randn(n, device='cuda') for _ in range(3))
    c = _native.StepCtl(lr=0.1, mu=0.5, c_wd=8e-5, c_noise=0.3, n_train=50000.0, flags=0x1 | 0x8 | 0x20, seed=1, step=0)     # NOISE | WD | ADVANCE
    ctl = torch.frombuffer(bytearray(bytes(c)), dtype=torch.uint8).cuda()
    for _ in range(20):
        K.sgmcmc_step_ctl(th, g, m, ctl)               # advances its own control block
    note('k_sgmcmc_step_ctl', 'K1 workload size (control block)', 20 * n, elements=n)
    # K1 at 2^26
    n = 1 << 26
    th, g, m = (torch.randn(n, device='cuda') for _ in range(3))
    for k in range(10):
        K.sgmcmc_step(th, g, m, lr=0.1, mu=0.5, c_wd=8e-5, c_noise=0.3, n_train=50000.0, flags=0x1 | 0x8, seed=1, step=k)
    note('k_sgmcmc_step<', 'K1 2^26 elements (SGHMC + Philox)', 20 * n, elements=n)
    # K2 / K3 at the WideResNet-28-10 arena size
    n = 36546980 + (-36546980) % 4
    out = torch.empty(n, device='cuda')
    m[:n].abs_().add_(th[:n] * th[:n])
    sd = torch.empty(n, device='cuda')
    for k in range(10):
        K.swag_std(sd, th[:n], m[:n], var_clamp=1e-30, scale=1.0)
    note('k_swag_std_v', 'K3 standard deviation, once per ensemble (WideResNet-28-10)', 12 * n, elements=n)
    for k in range(10):
        K.swag_draw_std(out, th[:n], sd, seed=3, draw=k)
    note('k_swag_draw_std_v', 'K3 WideResNet-28-10 member draw (Philox, std stored)', 12 * n, elements=n)
    for k in range(10):
        K.swag_draw(out, th[:n], m[:n], var_clamp=1e-30, scale=1.0, seed=3, draw=k)
    note('k_swag_draw_v', 'K3 fused single-draw form (Philox)', 12 * n, elements=n)
    del sd
    for k in range(10):
        K.swag_collect(th[:n], m[:n], g[:n], decay=0.75, denom=4.0)
    note('k_swag_collect', 'K2 WideResNet-28-10 moment update', 20 * n, elements=n)
    del th, g, m, out
    # K1: 4 PreResNet-20 chains in one self-advancing multi-chain launch
    nc = 273408
    ths, gs, ms_ = (torch.randn(4, nc, device='cuda') for _ in range(3))
    blocks = b''.join(bytes(_native.StepCtl(lr=0.1, mu=0.5, c_wd=8e-5, c_noise=0.3, n_train=50000.0, flags=0x1 | 0x8 | 0x20, seed=1 + k, step=0))
                      for k in range(4))
    ctl4 = torch.frombuffer(bytearray(blocks), dtype=torch.uint8).cuda()
    for _ in range(20):
        K.sgmcmc_step_multi(ths, gs, ms_, ctl4)
    manifest.append(dict(pattern='k_sgmcmc_step_ctl', label='K1 4 chains x 273,408 in one launch', algorithmic_bytes_per_launch=20 * nc * 4,
                         blocks=4 * ((nc // 4 + 1023) // 1024), wg=1024, elements_total=4 * nc))     # multi-chain launches: 1,024-thread workgroups
    del ths, gs, ms_
    # K4 in the launch forms inference/hmc.py issues, at PreResNet-164's size and at 2^26
    ws, acc = torch.zeros(2048, device='cuda'), torch.zeros(1, device='cuda')
    for nn, tag in ((1726400, '1.73M'), (1 << 26, '2^26')):
        t4, p4, g4 = (torch.randn(nn, device='cuda') for _ in range(3))
        for _ in range(10):
            K.leapfrog(t4, p4, g4, kick_coef=1e-4, step_size=2e-4, inv_mass=1.0, flags=0x3)
        note('k_leapfrog_v', f'K4 fused kick+drift {tag}', 20 * nn, elements=nn)
        for _ in range(10):
            K.leapfrog(None, p4, g4, kick_coef=-1e-4, step_size=2e-4, inv_mass=1.0, flags=0x1, kinetic_out=acc, ws=ws)
        manifest.append(dict(pattern='k_leapfrog<', label=f'K4 kick + kinetic-energy reduction {tag}', algorithmic_bytes_per_launch=12 * nn,
                             elements_total=nn, grid_threads=min((nn // 4 + 255) // 256, 2048) * 256))
        del t4, p4, g4
    # K5 at the shapes the tasks feed it
    for (S, B, C) in ((50, 10000, 10), (20, 9984, 10), (30, 10000, 100)):      # 9,984 rows: a grid of its own in the counter CSVs
        z = torch.randn(S, B, C, device='cuda') * 3
        p, e = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
        for _ in range(10):
            K.bma_accumulate(z, p, e, one_minus_gamma=0.9999, gamma_over_c=1e-4 / C, smoothed=False)
        grid = (B + 15) // 16 if C <= 16 else (B + 3) // 4
        # (C > 16: the lane-group kernel; bma_form in ursa_kernels.hip gives a test-set sized call with 8 classes per lane
        #  two member ranges per row group = 2 waves per block)
        wg = 64 * (4 if S >= 12 else 2 if S >= 5 else 1) if C <= 16 else (128 if (B + 3) // 4 >= 2048 and S >= 4 and 64 < C <= 128 else 256)
        note('k_bma_', f'K5 S={S} B={B} C={C}', 4 * S * B * C + 8 * B * (C + 1), shape=[S, B, C], blocks=grid, wg=wg)
    del z, p, e
if not K6_ONLY:
    # K7 / K8 at the workload's first-stage layer (16 channels, 32 x 32, batch 128): 8 MB operands; bytes = operands read once +
    # the result written once (K7's first launch also writes its 4.7 MB of K-sliced partial sums, read back by the second)
    cx, cdy = torch.randn(128, 16, 32, 32, device='cuda'), torch.randn(128, 16, 32, 32, device='cuda')
    cw, cy = torch.randn(16, 16, 3, 3, device='cuda') * 0.1, torch.empty(128, 16, 32, 32, device='cuda')
    cdw = torch.empty_like(cw)
    cws = torch.empty(K.conv_wgrad_ws_floats(cx.shape, 16, 3, 1), device='cuda')
    for _ in range(10):
        K.conv3x3(cx, cw, cy)
        K.conv3x3(cdy, cw, cy, flip=True)
        K.conv_wgrad(cx, cdy, cdw, cws, 1)
    manifest.append(dict(pattern='k_conv3x3<16, 16, 32, 8, 4, 0, 0, 0, 0>', label='K8 forward / input gradient 128x16x32x32', blocks=512, wg=256,
                         algorithmic_bytes_per_launch=4 * (2 * cx.numel() + cw.numel()), flops_per_launch=2 * 128 * 1024 * 16 * 16 * 9))
    manifest.append(dict(pattern='k_conv_wgrad<16, 16, 32', label='K7 first launch 128x16x32x32', blocks=512, wg=256,
                         algorithmic_bytes_per_launch=4 * (2 * cx.numel() + cw.numel()), form_bytes=4 * (2 * cx.numel() + cws.numel()),
                         flops_per_launch=2 * 128 * 1024 * 16 * 16 * 9))
    manifest.append(dict(pattern='k_conv_wgrad_reduce', label='K7 second launch (512 slices of 2,304 floats)', blocks=36, wg=256,
                         algorithmic_bytes_per_launch=4 * (cws.numel() + cw.numel())))
    # K10 at the same layer: the fused forward unit (BatchNorm + ReLU while staged, statistics of the result), the paired backward
    # launch (input gradient + BatchNorm-backward sums, weight gradient with the staged transform), K6's dx launch fed by them
    gam, bet = torch.rand(16, device='cuda') + 0.5, torch.randn(16, device='cuda') * 0.1
    xd = cx.double()
    ip = torch.stack([xd.sum((0, 2, 3)), (xd * xd).sum((0, 2, 3))], -1)[:, None, :].contiguous()
    save = torch.empty(4, 16, device='cuda')
    geo = K.preact_geometry(cx.shape, 16, bn=True)
    sc = torch.zeros(geo[1], dtype=torch.uint8, device='cuda')
    part = torch.empty(16, geo[0], 2, dtype=torch.float64, device='cuda')
    pb = torch.empty(16, K.preact_geometry(cdy.shape, 16, flip=True)[0], 2, dtype=torch.float64, device='cuda')
    cws2 = torch.empty(K.conv_wgrad_ws_floats(cx.shape, 16, 3, 1), device='cuda')
    g_, dx_, dgb = torch.empty_like(cx), torch.empty_like(cx), torch.empty(2, 16, device='cuda')
    for _ in range(10):
        K.preact_conv3x3(cx, cw, cy, part, sc, bn=(ip, gam, bet, None, None, save, 1e-5, 0.0))
        K.preact_conv3x3(cx, cw, cy, part, sc, bn=(ip, gam, bet, None, None, save, 1e-5, 0.0), add=cdy)
        K.preact_bwd_pair(cdy, cw, g_, cx, save, pb, cws2, 1)
        K.bn_bwd_dx(cx, g_, dx_, gam, save, pb, dgb[0], dgb[1], dz=cdy)
    fl = 2 * 128 * 1024 * 16 * 16 * 9
    manifest.append(dict(pattern='k_conv3x3<16, 16, 32, 8, 4, 0, 0, 1, 1>', label='K10 forward unit conv(relu(bn(x))) + statistics, 128x16x32x32', blocks=512, wg=256,
                         algorithmic_bytes_per_launch=4 * (2 * cx.numel() + cw.numel()), flops_per_launch=fl))
    manifest.append(dict(pattern='k_conv3x3<16, 16, 32, 8, 4, 0, 0, 1, 2>', label='K10 forward unit with the residual add, 128x16x32x32', blocks=512, wg=256,
                         algorithmic_bytes_per_launch=4 * (3 * cx.numel() + cw.numel()), flops_per_launch=fl))
    manifest.append(dict(pattern='k_bwd_pair<16, 16, 32', label='K10 paired backward launch (input gradient + weight gradient), 128x16x32x32', blocks=512 + cws2.numel() // 2304, wg=256,
                         algorithmic_bytes_per_launch=4 * (4 * cx.numel() + 2 * cw.numel()), form_bytes=4 * (4 * cx.numel() + cws2.numel()), flops_per_launch=2 * fl))
    manifest.append(dict(pattern='k_bn_bwd_dx<4, false, true, false, true>', label='K6 dx launch fed by K10 partial sums (+ shortcut gradient), 128x16x32x32', blocks=1024, wg=256,
                         algorithmic_bytes_per_launch=4 * 4 * cx.numel()))
    del cx, cdy, cy, cws, cws2, g_, dx_
# K6 relu(bn(x)): the two-launch form on a 268 MB activation (PreResNet-164's first stage at the HMC batch: beyond the
# Infinity Cache) and the one-pass form on an 84 MB one (WideResNet-28-10's last stage at 4x the batch)
for shape, one in K6_SHAPES:
    C = shape[1]
    x, dy = torch.randn(shape, device='cuda'), torch.randn(shape, device='cuda')
    y, dx = torch.empty_like(x), torch.empty_like(x)
    w, b = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    rm, rv, sm, si, dg, db = (torch.zeros(C, device='cuda') for _ in range(6))
    rv.fill_(1.0)
    wsb = torch.empty(_native.bn_ws_floats(C), device='cuda')
    for _ in range(10):
        K.bn_relu_forward(x, y, w, b, rm, rv, sm, si, wsb, eps=1e-5, momentum=0.1)
        K.bn_relu_backward(x, dy, dx, w, b, sm, si, dg, db, wsb)
        K.bn_relu_eval(x, y, w, b, rm, rv, eps=1e-5)
    e = x.numel()
    tag = 'x'.join(str(v) for v in shape)
    if one:
        manifest.append(dict(pattern='k_bn_fwd_one', label=f'K6 forward, one-pass form {tag}', algorithmic_bytes_per_launch=8 * e, blocks=C, wg=1024))
        manifest.append(dict(pattern='k_bn_bwd_one', label=f'K6 backward, one-pass form {tag}', algorithmic_bytes_per_launch=12 * e, blocks=C, wg=1024))
        S = 2
    else:
        S = min(64, -(-1024 // C))
        per = e // C // 4
        chunk = -(-(-(-per // S)) // 256) * 256
        S = -(-per // chunk)
        manifest.append(dict(pattern='k_bn_stats', label=f'K6 forward launch 1 (statistics) {tag}', algorithmic_bytes_per_launch=4 * e, blocks=S * C, wg=256))
        manifest.append(dict(pattern='k_bn_fwd_apply', label=f'K6 forward launch 2 (normalise + ReLU) {tag}', algorithmic_bytes_per_launch=8 * e, blocks=S * C, wg=256))
        manifest.append(dict(pattern='k_bn_bwd_reduce', label=f'K6 backward launch 1 (sums) {tag}', algorithmic_bytes_per_launch=8 * e, blocks=S * C, wg=256))
        manifest.append(dict(pattern='k_bn_bwd_dx', label=f'K6 backward launch 2 (dx) {tag}', algorithmic_bytes_per_launch=12 * e, blocks=S * C, wg=256))
    Se = min(64, -(-1024 // C))
    per = e // C // 4
    chunk = -(-(-(-per // Se)) // 256) * 256
    Se = -(-per // chunk)
    manifest.append(dict(pattern='k_bn_eval', label=f'K6 evaluation {tag}', algorithmic_bytes_per_launch=8 * e, blocks=Se * C, wg=256))
    if not one and e * 4 >= _native.BN_HELD_MIN_BYTES_FWD:
        # the held form of the same layer (one launch per direction, chunks held in registers): its HBM traffic must be the
        # ALGORITHMIC bytes - x read once (forward), x and dy read once (backward)
        wsz = torch.zeros(_native.bn_ws_floats(C), device='cuda')
        for _ in range(10):
            K.bn_relu_forward(x, y, w, b, rm, rv, sm, si, wsz, eps=1e-5, momentum=0.1, held=True)
            K.bn_relu_backward(x, dy, dx, w, b, sm, si, dg, db, wsz, held=True)
        per = e // C // 4

        def held_blocks(block, slots):            # csrc/ursa_bn.hip bn_held_plan: registers + LDS-held float4 per thread
            S = -(-per // (block * slots))        # (the largest shape; the forward's round-fill rule does not bite at these sizes)
            chunk = -(-(-(-per // S)) // block) * block
            return -(-per // chunk) * C
        manifest.append(dict(pattern='k_bn_fwd_held', label=f'K6 forward, held form (1 launch) {tag}', algorithmic_bytes_per_launch=8 * e,
                             blocks=held_blocks(512, 32 + 9), wg=512))
        manifest.append(dict(pattern='k_bn_bwd_held', label=f'K6 backward, held form (1 launch) {tag}', algorithmic_bytes_per_launch=12 * e,
                             blocks=held_blocks(512, 8 + 4), wg=512))
    del x, dy, y, dx
torch.cuda.synchronize()
json.dump(manifest, open(sys.argv[1], 'w'), indent=1)
