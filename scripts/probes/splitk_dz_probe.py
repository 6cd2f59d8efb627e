"""Round 6 probe: the two latency-bound input-gradient GEMMs of a G step (dz = dy W with M = batch, N = 128, K = 16384 / 7168: hipBLASLt picks a
tile with no split-K, 74 + 27 us) against a split-K formulation through torch.bmm + sum."""
import torch
B = 64
for K in (16384, 7168):
    dy = torch.randn(B, K, device="cuda")
    W = torch.randn(K, 128, device="cuda")
    def t(f, n=50):
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    ref = dy @ W
    print(K, "mm %.1f us" % t(lambda: dy @ W))
    for S in (8, 16, 32, 64):
        f = lambda: torch.bmm(dy.view(B, S, K // S).transpose(0, 1), W.view(S, K // S, 128)).sum(0)
        err = float((f() - ref).abs().max() / ref.abs().max())
        print(K, "split", S, "%.1f us" % t(f), "rel err %.1e" % err)
