"""VERDICT r4 "weak" 10: bench.py's CPU baselines are PORTS (`"kind": "port"`: oracle/p_oracle.c, oracle/gan_cpu.py) because
/root/reference does not exist on the GPU box.  This script times the REAL reference beside the ports on ONE host -- the build
container, where the reference tree is -- on the same synthetic tensors, so that the ports' fairness as stand-ins can be read off:

    python -O scripts/cpu_ref_vs_port.py > profiles/r05_cpu_reference_vs_port.txt        (-O: shim S0 of oracle/ref_harness.py)

P: 8 clouds of 2048 points -> 128^3, forward + backward through the reference's own functions (oracle/ref_harness.py:
ref_forward_stages + ref_supervised_loss, autograd) against oracle/p_oracle.c (forward + analytic backward, 1 thread).
G: the reference's models.gan / utils.losses cycle (1 G step + 2 D steps, Adam(0, 0.9)) at batch 2, 256^2, against oracle/gan_cpu.py
running the same cycle from the same weights -- both on the same thread counts."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oracle import ref_harness as rh, p_oracle as po   # noqa: E402


def cpu_model():
    for ln in open("/proc/cpuinfo"):
        if ln.startswith("model name"):
            return ln.split(":", 1)[1].strip()
    return "?"


def time_it(fn, min_s=4.0, max_n=20):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > min_s or n >= max_n:
            return el / n, n


def proj(threads):
    B, N, S = 8, 2048, 128
    rs = np.random.RandomState(3)
    pc = ((rs.rand(B, N, 3) - 0.5) * 0.7).astype(np.float32)
    q = rs.randn(B, 4).astype(np.float32)
    sc = (1 / (1 + np.exp(-rs.randn(B, 1)))).astype(np.float32)
    mask = (rs.rand(B, 2 * S, 2 * S) > 0.5).astype(np.float32)
    taps = po.taps(3.0, 21, False)
    torch.set_num_threads(threads)

    def ref():
        tp, tq, ts = (torch.from_numpy(a).requires_grad_() for a in (pc, q, sc))
        out = rh.ref_forward_stages(tp, tq, ts, S=S)
        rh.ref_supervised_loss(out["proj"], torch.from_numpy(mask)).backward()

    def port():
        p = po.forward(pc, q, sc, S, taps)
        po.backward(pc, q, sc, po.sup_loss_bwd(p, mask), S, taps)

    tr, nr = time_it(ref)
    tp_, np_ = time_it(port)
    return B / tr, nr, B / tp_, np_


def gan(threads):
    import argparse
    import gen_golden_g as gg
    from oracle import gan_cpu as gc
    ref_gan, GANLoss = gg.import_reference()
    args = gg.make_args(texture_resolution=256)
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        G = ref_gan.Generator(args, 64, symmetric=True, mesh_head=True)
        D = ref_gan.MultiScaleDiscriminator(args, 4)
    crit = GANLoss("hinge", tensor=torch.FloatTensor)
    B, R = 2, 256
    z, c, x_tex, x_alpha, x_mesh = gg.make_inputs(5, B, R, 200)
    og = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.0, 0.9))
    od = torch.optim.Adam(D.parameters(), lr=4e-4, betas=(0.0, 0.9))
    wg, wd = gc.Weights(G.state_dict()), gc.Weights(D.state_dict())
    pg = {k: v for k, v in wg.store.items() if v.requires_grad}
    pd = {k: v for k, v in wd.store.items() if v.requires_grad}
    sg, sd, step = {}, {}, [0, 0]
    torch.set_num_threads(threads)

    def ref_cycle():
        G.train(); D.train()
        og.zero_grad(); od.zero_grad()
        pt, pm = G(z, c, None)
        disc, mask = D(torch.cat((pt * x_alpha, x_alpha), 1), pm, c, None)
        crit(disc, True, for_discriminator=False, mask=mask, weight=None).mean().backward()
        og.step()
        for _ in range(2):
            od.zero_grad()
            with torch.no_grad():
                ft, fm = G(z, c, None)
                xc = torch.cat((torch.cat((ft * x_alpha, x_alpha), 1), torch.cat((x_tex, x_alpha), 1)), 0)
            d2, m2 = D(xc, torch.cat((fm, x_mesh), 0), torch.cat((c, c), 0), None)
            lf = crit([t[:B] for t in d2], False, for_discriminator=True, mask=[t[:B] for t in m2], weight=None)
            lr = crit([t[B:] for t in d2], True, for_discriminator=True, mask=[t[B:] for t in m2], weight=None)
            (lf + lr).mean().backward()
            od.step()

    def port_cycle():
        wg.zero_grad(); wd.zero_grad()
        gc.g_step(wg, wd, args, z, c, x_alpha)[0].mean().backward()
        step[0] += 1
        gc.adam_step(pg, wg.grads(), sg, 1e-4, step[0])
        for _ in range(2):
            wd.zero_grad()
            lf, lr, _ = gc.d_step(wg, wd, args, z, c, x_tex, x_alpha, x_mesh)
            (lf + lr).mean().backward()
            step[1] += 1
            gc.adam_step(pd, wd.grads(), sd, 4e-4, step[1])

    tr, nr = time_it(ref_cycle, min_s=8.0, max_n=6)
    tp_, np_ = time_it(port_cycle, min_s=8.0, max_n=6)
    return 3 * B / tr, nr, 3 * B / tp_, np_


if __name__ == "__main__":
    ncpu = os.cpu_count()
    print(f"# host: {cpu_model()}, {ncpu} hardware threads (the build container, NOT the GPU box); torch {torch.__version__} CPU, fp32")
    print("# samples/s, the REAL reference (/root/reference/code executed) beside bench.py's ports, same tensors, same thread count")
    js = {"host": cpu_model(), "hardware_threads": ncpu, "where": "build container (the reference tree does not exist on the GPU box)",
          "how": "python -O scripts/cpu_ref_vs_port.py --json profiles/cpu_port_over_reference.json", "proj": {}, "gan": {}}
    for thr in (1, ncpu):
        r, nr, p, np_ = proj(thr)
        js["proj"][str(thr)] = {"reference_samples_per_s": r, "port_samples_per_s": p, "port_over_reference": p / r,
                                "note": "port = oracle/p_oracle.c, scalar, always 1 thread"}
        print(f"projection fwd+bwd, 8 x 2048 pts -> 128^3, {thr:2d} thread(s): reference {r:8.2f} ({nr} runs)   port oracle/p_oracle.c (always 1 thread) "
              f"{p:8.2f} ({np_} runs)   port / reference = {p / r:.2f}")
    for thr in (1, ncpu):
        r, nr, p, np_ = gan(thr)
        js["gan"][str(thr)] = {"reference_samples_per_s": r, "port_samples_per_s": p, "port_over_reference": p / r,
                               "note": "port = oracle/gan_cpu.py on the same thread count"}
        print(f"GAN cycle (1 G + 2 D steps, Adam), batch 2, 256^2, {thr:2d} thread(s): reference {r:8.3f} ({nr} cycles)   port oracle/gan_cpu.py "
              f"{p:8.3f} ({np_} cycles)   port / reference = {p / r:.2f}")
    if "--json" in sys.argv:
        import json
        json.dump(js, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
